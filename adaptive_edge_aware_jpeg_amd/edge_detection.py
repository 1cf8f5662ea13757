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
        """Returns the float32 {0,1} edge map.  The hyper-parameters are compile-time constants of the
        kernels (no caller of the reference overrides them, jpeg.py:376); other values raise."""
        if not isinstance(img, np.ndarray):
            raise TypeError("Input must be a numpy array.")
        if img.ndim != 2:
            raise ValueError("Input array must be a 2D.")
        given = (aperture_size, use_L2_gradient, canny_low_ratio, canny_high_ratio, clahe_clip_limit,
                 tuple(clahe_tile_grid), bilateral_diameter, bilateral_sigma_color, bilateral_sigma_space, gaussian_kernel)
        if given != (3, True, 0.10, 0.30, 0.75, (4, 4), 5, 75, 75, 3):
            raise NotImplementedError("the HIP Canny chain is built for the reference's default hyper-parameters only")
        ctx = get_context()
        t = ctx.torch
        H, W = img.shape
        plane = ctx.to_device(img, t.float32)
        edge = ctx.empty((H, W), t.uint8)
        stages = ctx.empty((5, H, W), t.uint8) if return_stages else None
        thr = ctx.empty((2,), t.int32) if return_stages else None
        nbytes = ctx.lib.aej_canny_workspace_bytes(H, W)
        ws = ctx.workspace(nbytes)
        ctx.check(ctx.lib.aej_canny(ctx.handle, plane.data_ptr(), H, W, edge.data_ptr(),
                                    stages.data_ptr() if return_stages else None,
                                    thr.data_ptr() if return_stages else None, ws.data_ptr(), ctypes.c_uint64(nbytes)))
        out = edge.cpu().numpy().astype(np.float32)
        if return_stages:
            return out, stages.cpu().numpy(), tuple(int(v) for v in thr.cpu().numpy())
        return out
