"""Load the *pure-Python* parts of the reference from /root/reference/src for fixture generation.

Only used by the make_golden*.py scripts, in the build container (the reference never travels
to the GPU box and nothing under tests/ reads it at test time -- tests read the committed .npz
/ .json fixtures only).

The reference needs cv2 / numba / imageio, none of which are installed (ordinary
ModuleNotFoundError).  We register import stand-ins:

* ``numba``: ``njit`` returns the undecorated Python function, ``prange`` is ``range``.
  Scalar helper functions (``_pq_eotf`` / ``_pq_inverse_eotf``) are additionally wrapped so that
  numpy-float32 scalar arguments are widened to float64 and the result is a numpy float64: that
  is numba's typing (float32 op float64-constant -> float64), which NumPy-2 "weak scalar"
  promotion would otherwise not reproduce (SURVEY.md section 8c caveat).
* ``cv2``: an empty module unless the caller passes an object implementing the handful of calls
  (used by make_golden_compress.py, where the oracle provides them).
* ``imageio.v3``: empty module.
* ``image``: synthetic package that loads only image/image.py (image/__init__.py imports the
  LPIPS metric whose constructor fetches network weights).
"""
import importlib.util
import sys
import types

import numpy as np

REF_SRC = "/root/reference/src"


def _numba_standin():
    nb = types.ModuleType("numba")

    def njit(*args, **kwargs):
        def wrap(fn):
            if fn.__name__ in ("_pq_eotf", "_pq_inverse_eotf"):
                def widened(x, **kw):
                    return np.float64(fn(np.float64(x), **kw))
                widened.__name__ = fn.__name__
                return widened
            return fn
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return wrap(args[0])
        return wrap

    nb.njit = njit
    nb.prange = range
    return nb


def load_reference(cv2_module=None):
    """Returns a dict of the reference modules (color, jpeg.quadtree, jpeg.jpeg, jpeg.utils, image)."""
    sys.modules["numba"] = _numba_standin()
    sys.modules["cv2"] = cv2_module if cv2_module is not None else types.ModuleType("cv2")
    iio_pkg = types.ModuleType("imageio")
    iio_v3 = types.ModuleType("imageio.v3")
    iio_pkg.v3 = iio_v3
    sys.modules["imageio"] = iio_pkg
    sys.modules["imageio.v3"] = iio_v3

    # synthetic 'image' package: only image/image.py
    pkg = types.ModuleType("image")
    pkg.__path__ = [REF_SRC + "/image"]
    sys.modules["image"] = pkg
    spec = importlib.util.spec_from_file_location("image.image", REF_SRC + "/image/image.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["image.image"] = mod
    spec.loader.exec_module(mod)
    pkg.Image = mod.Image

    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import color  # noqa
    import jpeg.utils  # noqa
    import jpeg.quadtree  # noqa
    import jpeg.jpeg  # noqa
    import jpeg.edge_detection  # noqa
    return {
        "color": sys.modules["color"],
        "quadtree": sys.modules["jpeg.quadtree"],
        "jpeg": sys.modules["jpeg.jpeg"],
        "utils": sys.modules["jpeg.utils"],
        "edge": sys.modules["jpeg.edge_detection"],
        "image": pkg,
    }
