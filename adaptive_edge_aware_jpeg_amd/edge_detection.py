"""``EdgeDetection.canny`` (src/jpeg/edge_detection.py:25-86) on the GPU via ``aej_canny``."""
import ctypes
from typing import Tuple

import numpy as np

from ._lib import get_context


class EdgeDetection:
    """A collection of edge detection algorithms."""

    @staticmethod
    def canny(
        img: np.ndarray,
        aperture_size: int = 3,
        use_L2_gradient: bool = True,
        canny_low_ratio: float = 0.10,
        canny_high_ratio: float = 0.30,
        clahe_clip_limit: float = 0.75,
        clahe_tile_grid: Tuple[int, int] = (4, 4),
        bilateral_diameter: int = 5,
        bilateral_sigma_color: int = 75,
        bilateral_sigma_space: int = 75,
        gaussian_kernel: int = 3,
        return_stages: bool = False,
    ) -> np.ndarray:
        """Returns the float32 {0,1} edge map.  The threshold ratios, the CLAHE clip limit, the bilateral sigmas and the
        gradient norm are run-time values of the kernels (``aej_set_canny_params``); the stencil shapes -- ``aperture_size`` 3,
        ``clahe_tile_grid`` (4, 4), ``bilateral_diameter`` 5, ``gaussian_kernel`` 3, the values the reference's only caller uses
        (jpeg.py:376) -- are structural, other values raise."""
        if not isinstance(img, np.ndarray):
            raise TypeError("Input must be a numpy array.")
        if img.ndim != 2:
            raise ValueError("Input array must be a 2D.")
        shape = (aperture_size, tuple(clahe_tile_grid), bilateral_diameter, gaussian_kernel)
        if shape != (3, (4, 4), 5, 3):
            raise NotImplementedError("the HIP Canny chain is built for aperture_size=3, clahe_tile_grid=(4, 4), bilateral_diameter=5, "
                                      f"gaussian_kernel=3 (got {shape})")
        params = (canny_low_ratio, canny_high_ratio, clahe_clip_limit, bilateral_sigma_color, bilateral_sigma_space, bool(use_L2_gradient))
        default = params == (0.10, 0.30, 0.75, 75, 75, True)
        ctx = get_context()
        t = ctx.torch
        H, W = img.shape
        plane = ctx.to_device(img, t.float32)
        edge = ctx.empty((H, W), t.uint8)
        stages = ctx.empty((5, H, W), t.uint8) if return_stages else None
        thr = ctx.empty((2,), t.int32) if return_stages else None
        nbytes = ctx.lib.aej_canny_workspace_bytes(H, W)
        ws = ctx.workspace(nbytes)
        if not default:
            ctx.set_canny_params(params)
        try:
            ctx.check(ctx.lib.aej_canny(ctx.handle, plane.data_ptr(), H, W, edge.data_ptr(),
                                        stages.data_ptr() if return_stages else None,
                                        thr.data_ptr() if return_stages else None, ws.data_ptr(), ctypes.c_uint64(nbytes)))
        finally:
            if not default:
                ctx.set_canny_params(None)          # the context is shared with Jpeg, whose Canny stage uses the defaults
        out = edge.cpu().numpy().astype(np.float32)
        if return_stages:
            return out, stages.cpu().numpy(), tuple(int(v) for v in thr.cpu().numpy())
        return out
