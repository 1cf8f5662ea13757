"""Generate color_xyz.npz: convert("sRGB", "XYZ", x) and convert("XYZ", "sRGB", y) executed from the reference's own
modules (conversion.py:119-124 -> xyz.py:63-64, 83-84), with numba's float64 typing of the sRGB curve emulated exactly as
make_golden.py / make_golden_decode.py do (float64 copies of the float32 data).  Build container only; commit the output.

    python tests/golden/make_golden_xyz.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference  # noqa: E402


def main():
    ref = load_reference()
    color = ref["color"]
    from color.xyz import XYZ
    from color.common import _linear_rgb_to_srgb
    fwd = np.load(os.path.join(HERE, "color_forward.npz"))
    u8 = fwd["rgb_u8"]
    x32 = u8.astype(np.float32) / 255.0                      # Image.load's values (image.py:80)
    xyz = color.convert("sRGB", "XYZ", x32.astype(np.float64))   # float64 copy: numba evaluates the sRGB curve in float64
    assert xyz.dtype == np.float32
    rng = np.random.default_rng(13)
    y = xyz + rng.normal(0, 0.002, xyz.shape).astype(np.float32)
    lin = np.dot(y, XYZ.M_XYZ_TO_LINEAR_RGB_T)               # XYZ.xyz_to_srgb, first statement (float32 sgemm)
    assert lin.dtype == np.float32
    back = _linear_rgb_to_srgb(lin.astype(np.float64))       # second statement with numba's typing
    assert back.dtype == np.float32
    norm = color.apply_normalization("XYZ", xyz, False).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "color_xyz.npz"), rgb_u8=u8, XYZ=xyz, XYZ_norm=norm, XYZ_in=y, sRGB=back)
    print("xyz done", xyz.shape, float(np.abs(back - x32).max()))


if __name__ == "__main__":
    main()
