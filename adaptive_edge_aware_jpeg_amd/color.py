"""``color.convert`` / ``apply_normalization`` / ``get_color_spaces`` (src/color/conversion.py:86-157) on the GPU.

Forward transforms (sRGB -> space) run in the HIP kernels of csrc/color.hip through ``aej_color_convert``, inverse
transforms (decode path) in csrc/decode.hip through ``aej_color_convert_inverse``.
"""
import ctypes

import numpy as np

from ._lib import CONVERT_IDS, get_context

# MIDPOINTS / SCALE_FACTORS exactly as the reference declares them (float32 of these literals):
# ycbcr.py:41-42, ycocg.py:41-42,62-63, oklab.py:51-52, ictcp.py:162-163, icacb.py:162-163, jzazbz.py:211-212,
# xyz.py:41-42
_NORM = {
    "YCbCr": ([0.5000000037252903, 7.450580596923828e-09, 0.0], [253.99999810755253, 254.000003784895, 254.0]),
    "YCoCg": ([0.5, 0.0, 0.0], [254.0, 254.0, 254.0]),
    "YCoCg-R": ([0.5, 0.0, 0.0], [254.0, 127.0, 127.0]),
    "OKLAB": ([0.4999999, 0.021152213, -0.056563325], [254.00005, 497.9055, 497.94604]),
    "ICtCp": ([0.07497266, -0.0008235276, 0.023989676], [1693.9674, 1133.9044, 1694.004]),
    "ICaCb": ([0.07498085, 0.02180194, -0.018250957], [1693.7823, 1838.5665, 1330.3855]),
    "JzAzBz": ([0.0087900255, 0.00048353244, -0.0020741792], [14448.194, 7590.505, 5552.201]),
    "XYZ": ([0.47523502, 0.50000006, 0.544415], [267.2362, 253.99997, 233.27792]),
}
MIDPOINTS = {k: np.array(v[0], dtype=np.float32) for k, v in _NORM.items()}
SCALE_FACTORS = {k: np.array(v[1], dtype=np.float32) for k, v in _NORM.items()}

_ALL_SPACES = ("sRGB", "ICaCb", "ICtCp", "JzAzBz", "OKLAB", "YCbCr", "XYZ", "YCoCg", "YCoCg-R")


def get_color_spaces():
    """conversion.py:86-93"""
    return list(set(_ALL_SPACES) - {"sRGB"} - {"XYZ"})


def _check(data):
    if not isinstance(data, np.ndarray):
        raise TypeError("Data input must be a numpy array.")
    if data.ndim != 2 or data.shape[1] != 3:
        raise ValueError("Data input array must be a 2D with 3 channels.")


def convert(from_space: str, to_space: str, data: np.ndarray) -> np.ndarray:
    """conversion.py:95-124.  Computes in float32 on the GPU: the codec always feeds float32 (image.py:80); float64 input
    (the reference's own colour unit test feeds a float64 grid, test_color_conversions.py:31-33) is rounded to float32 first
    and the result is float32, as the reference's transforms return."""
    _check(data)
    if from_space not in _ALL_SPACES or to_space not in _ALL_SPACES:
        raise ValueError("Invalid color space. Please check the available color spaces.")
    if from_space != "sRGB" and to_space != "sRGB":
        raise ValueError("One of the color spaces must be sRGB.")
    if from_space == "sRGB":
        if to_space == "sRGB":
            return None     # the reference's table holds None callables for sRGB and would raise; keep it inert
        ctx = get_context()
        t = ctx.torch
        x = ctx.to_device(data, t.float32)
        out = t.empty_like(x)
        ctx.check(ctx.lib.aej_color_convert(ctx.handle, CONVERT_IDS[to_space], x.data_ptr(), out.data_ptr(),
                                            ctypes.c_int64(x.shape[0])))
        return out.cpu().numpy()
    ctx = get_context()
    t = ctx.torch
    x = ctx.to_device(data, t.float32)
    out = t.empty_like(x)
    ctx.check(ctx.lib.aej_color_convert_inverse(ctx.handle, CONVERT_IDS[from_space], x.data_ptr(), out.data_ptr(),
                                                ctypes.c_int64(x.shape[0])))
    return out.cpu().numpy()


def apply_normalization(color_space: str, data: np.ndarray, inverse: bool) -> np.ndarray:
    """conversion.py:126-157 / common.py:161-189.  Elementwise on the host for the stand-alone API; the encode
    path applies it inside the fused colour kernel."""
    _check(data)
    if color_space not in _ALL_SPACES or color_space == "sRGB":
        raise ValueError("Invalid color space. Please check the available color spaces.")
    mid, sc = MIDPOINTS[color_space], SCALE_FACTORS[color_space]
    if inverse:
        return data / sc + mid
    return (data - mid) * sc
