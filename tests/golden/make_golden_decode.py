"""Reference-derived golden vectors for the DECODE path (build container only).

    python tests/golden/make_golden_decode.py

color_inverse.npz   convert(space, "sRGB", y) for the 7 spaces, executed from the reference's modules (numba typing
                    emulated as in make_golden.py: float64 copies / np.float64 scalars where numba would promote)
decode_cases.json   sha256 of the float32 image returned by the REFERENCE's own Jpeg.decompress (src/jpeg/jpeg.py:274-297)
                    for the committed .ajpg fixtures, with the oracle standing in for cv2.idct / cv2.resize, plus a
                    64x64 corner of each result (decode_cases.npz) for diagnostics
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from _ref_loader import load_reference  # noqa: E402
from make_golden_compress import make_cv2  # noqa: E402


def make_cv2_decode():
    cv = make_cv2()
    base_resize = cv.resize

    def resize(src, dsize, interpolation=1):
        src = np.ascontiguousarray(src, dtype=np.float32)
        if interpolation == cv.INTER_LINEAR and src.shape != (8, 8):
            return O.upsample_linear(src, int(dsize[1]), int(dsize[0]))
        return base_resize(src, dsize, interpolation)

    def idct(block):
        block = np.ascontiguousarray(block, dtype=np.float32)
        s = block.shape[0]
        # oracle.blocks_decode with unit quantisers, identity zigzag, mid 0, scale 1 == the bare IDCT contract
        lib = O.lib()
        import ctypes
        import math
        coeffs = None
        # blocks_decode takes integer coefficients; call the C routine on float data through a tiny detour:
        # Y = (float)(c * 1) needs integers, so use the dedicated entry below instead
        out = np.empty((s, s), np.float32)
        D = O.dct_matrix(s)
        lib.orc_idct_block(D.ctypes.data_as(ctypes.c_void_p), block.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), s)
        return out

    cv.resize, cv.idct = resize, idct
    return cv


def main():
    ref = load_reference(make_cv2_decode())
    color = ref["color"]
    # ---- inverse colour
    fwd = np.load(os.path.join(HERE, "color_forward.npz"))
    rng = np.random.default_rng(11)
    out = {}
    for sp in ("YCbCr", "YCoCg", "YCoCg-R"):
        y = fwd[sp] + rng.normal(0, 0.01, fwd[sp].shape).astype(np.float32)          # decoded values are perturbed
        out[sp + "_in"] = y
        out[sp] = color.convert(sp, "sRGB", y)
        assert out[sp].dtype == np.float32
    from color.xyz import XYZ
    from color.common import _linear_rgb_to_srgb
    from color.oklab import OKLAB
    from color import ictcp as ict_mod, icacb as ica_mod, jzazbz as jz_mod

    def xyz_to_srgb(xyz):          # XYZ.xyz_to_srgb with numba's typing of _linear_rgb_to_srgb (float64 arithmetic)
        lin = np.dot(xyz, XYZ.M_XYZ_TO_LINEAR_RGB_T)
        assert lin.dtype == np.float32
        return _linear_rgb_to_srgb(lin.astype(np.float64))

    y = fwd["OKLAB"] + rng.normal(0, 0.002, fwd["OKLAB"].shape).astype(np.float32)
    lms_p = np.dot(y, OKLAB.M_LAB_TO_LMS_P_T)
    lms = np.power(lms_p, 3)
    out["OKLAB_in"], out["OKLAB"] = y, xyz_to_srgb(np.dot(lms, OKLAB.M_LMS_TO_XYZ_T))
    y = fwd["ICtCp"] + rng.normal(0, 0.0005, fwd["ICtCp"].shape).astype(np.float32)
    out["ICtCp_in"], out["ICtCp"] = y, xyz_to_srgb(ict_mod._ictcp_to_xyz(y, ict_mod.ICtCp.M_LMS_TO_XYZ, ict_mod.ICtCp.M_ICTCP_TO_LMS_P))
    y = fwd["ICaCb"] + rng.normal(0, 0.0005, fwd["ICaCb"].shape).astype(np.float32)
    out["ICaCb_in"], out["ICaCb"] = y, xyz_to_srgb(ica_mod._icacb_to_xyz(y, ica_mod.ICaCb.M_RGB_BAR_TO_XYZ, ica_mod.ICaCb.M_ICACB_TO_RGB_P))
    J = jz_mod.JzAzBz
    y = fwd["JzAzBz"] + rng.normal(0, 0.00005, fwd["JzAzBz"].shape).astype(np.float32)
    xyz = jz_mod._jzazbz_to_xyz(y, np.float64(J.B), np.float64(J.G), np.float64(J.D), np.float64(J.D0), np.float64(J.P),
                                J.M_LMS_TO_XYZ, J.M_IZAZBZ_TO_LMS_P)
    out["JzAzBz_in"], out["JzAzBz"] = y, xyz_to_srgb(xyz)
    for k, v in out.items():
        assert v.dtype == np.float32, (k, v.dtype)
    np.savez_compressed(os.path.join(HERE, "color_inverse.npz"), **out)
    print("inverse colour done")

    # ---- whole decompress through the reference
    Jpeg, Settings = ref["jpeg"].Jpeg, ref["jpeg"].JpegCompressionSettings
    meta, crops = {}, {}
    for name in ("lena_ycbcr_8_8_q50", "lena_default_ycocg_4_64", "crop_ycbcr_4_64"):
        data = open(os.path.join(HERE, name + ".ajpg"), "rb").read()
        img = Jpeg(Settings()).decompress(data)          # fresh default codec, as test_compression_speed.py:73 does
        arr = np.ascontiguousarray(img.data, dtype=np.float32)
        meta[name] = {"shape": list(arr.shape), "sha256": hashlib.sha256(arr.tobytes()).hexdigest(), "extension": img.extension}
        crops[name] = arr[:64, :64].copy()
        print(name, arr.shape, meta[name]["sha256"][:16])
    json.dump(meta, open(os.path.join(HERE, "decode_cases.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "decode_cases.npz"), **crops)


if __name__ == "__main__":
    main()
