"""Generate reference-derived golden vectors (build container only; run once, commit the outputs).

    python tests/golden/make_golden.py

Writes into tests/golden/:
  quadtree_cases.npz     QuadTree(edge,max,min).get_leaves_and_states() for random edge maps
  zigzag.npz             Jpeg._zigzag_ordering(s), s = 2..256
  quality.json           Jpeg._get_quality_factor tables + largest_power_of_2 + _decode_leaf_sizes
  color_forward.npz      convert("sRGB", space, x) for the 7 spaces on a fixed colour set
  color_constants.json   MIDPOINTS / SCALE_FACTORS / matrices as float32 hex strings

Everything here is produced by executing the reference's own Python (see _ref_loader.py for the
import stand-ins); no reference source text is stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference  # noqa: E402

ref = load_reference()
QuadTree = ref["quadtree"].QuadTree
Jpeg = ref["jpeg"].Jpeg
JpegCompressionSettings = ref["jpeg"].JpegCompressionSettings
largest_power_of_2 = ref["utils"].largest_power_of_2
color = ref["color"]


def f32hex(a):
    a = np.asarray(a, dtype=np.float32)
    return [format(int(v), "08x") for v in a.ravel().view(np.uint32)]


# ---------------------------------------------------------------- quadtree
def quadtree_cases():
    rng = np.random.default_rng(20250718)
    shapes = [(1, 1), (1, 7), (2, 2), (3, 5), (8, 8), (16, 16), (17, 9), (31, 33), (64, 64),
              (65, 64), (100, 37), (128, 128), (150, 90), (129, 257), (200, 300), (256, 256),
              (67, 259), (33, 1), (4, 4), (5, 4), (96, 160), (270, 480)]
    ranges = [(4, 64), (4, 64), (8, 128), (2, 8), (4, 4), (8, 8), (4, 128), (16, 32), (2, 256), (4, 16)]
    out = {}
    n = 0
    for si, (h, w) in enumerate(shapes):
        for di, density in enumerate((0.0, 0.002, 0.02, 0.3)):
            mn, mx = ranges[(si * 4 + di) % len(ranges)]
            edge = (rng.random((h, w)) < density).astype(np.float32)
            if density > 0 and di == 2:
                # a few line-like structures, as Canny output would have
                y = rng.integers(0, h)
                edge[y, :] = 1.0
            qt = QuadTree(edge, max_size=mx, min_size=mn)
            leaves, states = qt.get_leaves_and_states()
            out[f"c{n}_edge"] = edge.astype(np.uint8)
            out[f"c{n}_params"] = np.array([mn, mx, qt.root.size], dtype=np.int32)
            out[f"c{n}_leaves"] = np.array([[l.x, l.y, l.size] for l in leaves], dtype=np.int32).reshape(-1, 3)
            out[f"c{n}_states"] = np.array([int(s, 2) for s in states], dtype=np.uint8)
            # round trip through the reference's own header decoder
            sizes = Jpeg._decode_leaf_sizes([int(s, 2) for s in states], qt.root.size)
            assert sizes == [l.size for l in leaves]
            n += 1
    out["n_cases"] = np.array([n], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "quadtree_cases.npz"), **out)
    print("quadtree cases:", n)


# ---------------------------------------------------------------- zigzag / quality
def zigzag_and_quality():
    zz = {f"s{s}": Jpeg._zigzag_ordering(s) for s in (2, 4, 8, 16, 32, 64, 128, 256)}
    np.savez_compressed(os.path.join(HERE, "zigzag.npz"), **zz)

    class Stub:
        pass
    q = {}
    for brange in [(4, 64), (4, 128), (2, 256), (8, 8), (4, 4), (8, 128), (16, 32), (4, 16), (2, 8)]:
        for qrange in [(40, 80), (20, 60), (50, 50), (75, 75), (1, 99), (10, 90), (60, 20)]:
            st = Stub()
            st.settings = JpegCompressionSettings("YCbCr", qrange, brange)
            sizes = []
            s = brange[0]
            while s <= brange[1]:
                sizes.append(s)
                s *= 2
            q[f"{brange[0]}-{brange[1]}|{qrange[0]}-{qrange[1]}"] = {
                str(s): Jpeg._get_quality_factor(st, s) for s in sizes}
    lp2 = {str(n): int(largest_power_of_2(n)) for n in
           [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 270,
            480, 511, 512, 513, 540, 960, 1023, 1024, 1025, 1080, 1920, 2048, 2160, 3840, 4096, 4320, 7680, 8192]}
    ratios = {k: np.asarray(v["downsampling_ratios"]).tolist()
              for k, v in JpegCompressionSettings.COLOR_SPACE_SETTINGS.items()}
    shapes = {}
    for sp in ratios:
        st = Stub()
        st.settings = JpegCompressionSettings(sp)
        for hw in [(512, 512), (1080, 1920), (2160, 3840), (4320, 7680), (768, 512), (333, 517)]:
            shapes[f"{sp}|{hw[0]}x{hw[1]}"] = np.asarray(
                Jpeg._compute_downsampled_shapes(st, np.array(hw))).tolist()
    defaults = JpegCompressionSettings()
    json.dump({
        "quality": q, "largest_power_of_2": lp2, "ratios": ratios, "layer_shapes": shapes,
        "defaults": {"color_space": defaults.color_space, "quality_range": list(defaults.quality_range),
                     "block_size_range": list(defaults.block_size_range)},
        "lum_table": JpegCompressionSettings.LUMINANCE_QUANTIZATION_MATRIX.astype(int).tolist(),
        "chrom_table": JpegCompressionSettings.CHROMINANCE_QUANTIZATION_MATRIX.astype(int).tolist(),
        "color_spaces": sorted(color.get_color_spaces()),
    }, open(os.path.join(HERE, "quality.json"), "w"), indent=1, sort_keys=True)
    print("zigzag + quality done")


# ---------------------------------------------------------------- colour
def colour():
    rng = np.random.default_rng(7)
    grid = np.array(np.meshgrid(np.arange(0, 256, 17), np.arange(0, 256, 17), np.arange(0, 256, 17),
                                indexing="ij")).reshape(3, -1).T          # 16^3 = 4096 colours incl. greys
    extra = rng.integers(0, 256, size=(4096, 3))
    u8 = np.concatenate([grid, extra]).astype(np.uint8)
    x32 = u8.astype(np.float32) / 255.0            # exactly what Image.load produces (image.py:80)
    out = {"rgb_u8": u8}
    for sp in ("YCbCr", "YCoCg", "YCoCg-R"):
        y = color.convert("sRGB", sp, x32)
        assert y.dtype == np.float32
        out[sp] = y
        out[sp + "_norm"] = color.apply_normalization(sp, y, False).astype(np.float32)
    # numba-typed spaces: feed float64 copies of the float32 data so that the scalar arithmetic
    # runs in float64 as numba would run it for float32 input (see _ref_loader docstring).
    x64 = x32.astype(np.float64)
    for sp in ("OKLAB", "ICtCp", "ICaCb"):
        y = color.convert("sRGB", sp, x64)
        assert y.dtype == np.float32, (sp, y.dtype)
        out[sp] = y
        out[sp + "_norm"] = color.apply_normalization(sp, y, False).astype(np.float32)

    # JzAzBz mixes Python-float scalars (b, g, d, d0, p) with the float32 XYZ row inside the numba kernel;
    # numba promotes float64-scalar op float32 to float64 whereas NumPy-2 treats Python floats as "weak"
    # and would stay in float32.  Passing the same scalars as np.float64 (strongly typed) makes the
    # reference kernel body evaluate with numba's types; Z_p = Z and M[i,2]*Z_p stay float32 as in numba.
    from color.xyz import XYZ
    from color import jzazbz as jz_mod
    xyz = XYZ.srgb_to_xyz(x64)
    assert xyz.dtype == np.float32
    J = jz_mod.JzAzBz
    y = jz_mod._xyz_to_jzazbz(xyz, np.float64(J.B), np.float64(J.G), np.float64(J.D), np.float64(J.D0),
                              np.float64(J.P), J.M_XYZ_TO_LMS, J.M_LMS_P_TO_IZAZBZ)
    assert y.dtype == np.float32
    out["JzAzBz"] = y
    out["JzAzBz_norm"] = color.apply_normalization("JzAzBz", y, False).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "color_forward.npz"), **out)

    consts = {}
    for sp, (fwd, inv, mid, sc) in color.conversion.COLOR_CLASSES.items():
        if mid is None:
            continue
        consts[sp] = {"mid": f32hex(mid), "scale": f32hex(sc)}
    from color.ycbcr import YCbCr
    from color.ycocg import YCoCg
    from color.oklab import OKLAB
    from color.xyz import XYZ
    from color.ictcp import ICtCp
    from color.icacb import ICaCb
    from color.jzazbz import JzAzBz
    mats = {
        "YCbCr.fwd": YCbCr.M_SRGB_TO_YCBCR, "YCbCr.inv": YCbCr.M_YCBCR_TO_SRGB,
        "YCoCg.fwd": YCoCg.M_SRGB_TO_YCOCG, "YCoCg.inv": YCoCg.M_YCOCG_TO_SRGB,
        "YCoCg-R.fwd": YCoCg.M_SRGB_TO_YCOCG_R, "YCoCg-R.inv": YCoCg.M_YCOCG_R_TO_SRGB,
        "XYZ.fwd": XYZ.M_LINEAR_RGB_TO_XYZ, "XYZ.inv": XYZ.M_XYZ_TO_LINEAR_RGB,
        "OKLAB.m1": OKLAB.M_XYZ_TO_LMS, "OKLAB.m2": OKLAB.M_LMS_P_TO_LAB,
        "OKLAB.m1inv": OKLAB.M_LMS_TO_XYZ, "OKLAB.m2inv": OKLAB.M_LAB_TO_LMS_P,
        "ICtCp.m1": ICtCp.M_XYZ_TO_LMS, "ICtCp.m2": ICtCp.M_LMS_P_TO_ICTCP,
        "ICtCp.m1inv": ICtCp.M_LMS_TO_XYZ, "ICtCp.m2inv": ICtCp.M_ICTCP_TO_LMS_P,
        "ICaCb.m1": ICaCb.M_XYZ_TO_RGB_BAR, "ICaCb.m2": ICaCb.M_RGB_P_TO_ICACB,
        "ICaCb.m1inv": ICaCb.M_RGB_BAR_TO_XYZ, "ICaCb.m2inv": ICaCb.M_ICACB_TO_RGB_P,
        "JzAzBz.m1": JzAzBz.M_XYZ_TO_LMS, "JzAzBz.m2": JzAzBz.M_LMS_P_TO_IZAZBZ,
        "JzAzBz.m1inv": JzAzBz.M_LMS_TO_XYZ, "JzAzBz.m2inv": JzAzBz.M_IZAZBZ_TO_LMS_P,
    }
    for k, m in mats.items():
        assert m.dtype == np.float32, (k, m.dtype)
        consts[k] = f32hex(m)
    consts["JzAzBz.scalars"] = {"B": JzAzBz.B, "G": JzAzBz.G, "D": JzAzBz.D, "D0": JzAzBz.D0, "P": JzAzBz.P}
    json.dump(consts, open(os.path.join(HERE, "color_constants.json"), "w"), indent=1, sort_keys=True)
    print("colour done")


if __name__ == "__main__":
    quadtree_cases()
    zigzag_and_quality()
    colour()
