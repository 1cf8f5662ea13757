"""CPU ORACLE (numpy + C) for the adaptive-JPEG encode hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does.  The arithmetic lives in aej_oracle.c (scalar, order-defined C); this
file holds the ctypes binding plus the tiny host-side tables the reference builds in Python
(zigzag order, quality law, quantisation matrices, layer shapes) and the orchestration of
``Jpeg.compress`` (src/jpeg/jpeg.py:240-272).  Parity status per stage: header of aej_oracle.c.

Citations are path:line under /root/reference/.
"""
import ctypes
import json
import math
import os
import subprocess
import zlib
from io import BytesIO

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_SRC = os.path.join(_HERE, "aej_oracle.c")

SPACES = {"YCbCr": 0, "YCoCg": 1, "YCoCg-R": 2, "OKLAB": 3, "ICtCp": 4, "ICaCb": 5, "JzAzBz": 6, "XYZ": 7}   # XYZ: colour conversion only

# jpeg.py:62-147 -- (rh, rw) per layer
RATIOS = {
    "ICaCb": [(1, 1), (1, 4), (1, 4)], "ICtCp": [(1, 1), (1, 4), (1, 4)],
    "JzAzBz": [(1, 1), (2, 2), (2, 2)], "OKLAB": [(1, 1), (2, 2), (2, 2)], "YCbCr": [(1, 1), (2, 2), (2, 2)],
    "YCoCg": [(1, 1), (2, 2), (2, 2)], "YCoCg-R": [(1, 1), (2, 2), (2, 2)],
}

# MIDPOINTS / SCALE_FACTORS: np.array([...python doubles...], dtype=np.float32) in each colour file
NORM = {
    "YCbCr": ([0.5000000037252903, 7.450580596923828e-09, 0.0], [253.99999810755253, 254.000003784895, 254.0]),  # ycbcr.py:41-42
    "YCoCg": ([0.5, 0.0, 0.0], [254.0, 254.0, 254.0]),                                                             # ycocg.py:41-42
    "YCoCg-R": ([0.5, 0.0, 0.0], [254.0, 127.0, 127.0]),                                                           # ycocg.py:62-63
    "OKLAB": ([0.4999999, 0.021152213, -0.056563325], [254.00005, 497.9055, 497.94604]),                         # oklab.py:51-52
    "ICtCp": ([0.07497266, -0.0008235276, 0.023989676], [1693.9674, 1133.9044, 1694.004]),                       # ictcp.py:162-163
    "ICaCb": ([0.07498085, 0.02180194, -0.018250957], [1693.7823, 1838.5665, 1330.3855]),                        # icacb.py:162-163
    "JzAzBz": ([0.0087900255, 0.00048353244, -0.0020741792], [14448.194, 7590.505, 5552.201]),                   # jzazbz.py:211-212
}

LUM = np.array([[16, 11, 10, 16, 24, 40, 51, 61], [12, 12, 14, 19, 26, 58, 60, 55], [14, 13, 16, 24, 40, 57, 69, 56],
                [14, 17, 22, 29, 51, 87, 80, 62], [18, 22, 37, 56, 68, 109, 103, 77], [24, 35, 55, 64, 81, 104, 113, 92],
                [49, 64, 78, 87, 103, 121, 120, 101], [72, 92, 95, 98, 112, 100, 103, 99]], dtype=np.float32)  # jpeg.py:40-49
CHR = np.array([[17, 18, 24, 47, 99, 99, 99, 99], [18, 21, 26, 66, 99, 99, 99, 99], [24, 26, 56, 99, 99, 99, 99, 99],
                [47, 66, 99, 99, 99, 99, 99, 99]] + [[99] * 8] * 4, dtype=np.float32)                          # jpeg.py:50-59


_SO_ASAN = os.path.join(_HERE, "liboracle_asan.so")


def build(force=False, sanitize=False):
    """Compile aej_oracle.c -> liboracle.so (gcc, no contraction, no fast-math).  sanitize=True builds the
    -fsanitize=address,undefined variant liboracle_asan.so (CPU tests only: tests/test_oracle_pins.py runs the whole oracle path
    under it in a child interpreter with libasan preloaded; select it with AEJ_ORACLE_SANITIZE=1)."""
    hdrs = [os.path.join(_HERE, h) for h in ("aej_inv_constants.h", "aej_pow_tables.h")]
    so = _SO_ASAN if sanitize else _SO
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [_SRC] + hdrs):
        extra = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"] if sanitize else ["-O2"]
        # compiled to a private temporary and renamed into place: several ranks of one node may build at once (bench.py), and a
        # reader must never dlopen a half-written file
        tmp = f"{so}.{os.getpid()}.tmp"
        try:
            subprocess.check_call(["gcc"] + extra + ["-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
                                   "-fvisibility=hidden", "-mfma", "-mavx2", "-I", _HERE, "-o", tmp, _SRC, "-lm"])
            os.replace(tmp, so)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        sanitize = os.environ.get("AEJ_ORACLE_SANITIZE") == "1"
        _lib = ctypes.CDLL(build(sanitize=sanitize))
        _lib.orc_root_size.restype = ctypes.c_int
        _lib.orc_leaf_positions.restype = ctypes.c_int64
        for name in ("orc_downsample", "orc_quadtree", "orc_blocks_encode", "orc_blocks_decode"):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def _p(a, t=ctypes.c_void_p):
    return a.ctypes.data_as(t)


# ------------------------------------------------------------------ stage wrappers
def color_forward(space, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32).reshape(-1, 3)
    out = np.empty_like(rgb)
    lib().orc_color_forward(ctypes.c_int(SPACES[space]), _p(rgb), _p(out), ctypes.c_int64(rgb.shape[0]))
    return out


def pow_third_array(x, independent=False):
    """x ** float32(1 / 3) in float32: the short sequence the OKLAB kernel shares, or (independent) the general float64 pow"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().orc_pow_third_array(_p(x), _p(out), ctypes.c_int64(x.size), ctypes.c_int(1 if independent else 0))
    return out


def set_oklab_independent_pow(on):
    """tests only: OKLAB's cube root through the general float64 pow (no code in common with the HIP kernel)"""
    lib().orc_set_oklab_independent_pow(ctypes.c_int(1 if on else 0))


def pow_array(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().orc_pow_array(_p(x), ctypes.c_double(y), _p(out), ctypes.c_int64(x.size))
    return out


def downsample(conv_hw3, ch, rh, rw):
    conv = np.ascontiguousarray(conv_hw3, dtype=np.float32)
    H, W, _ = conv.shape
    out = np.empty((H // rh, W // rw), dtype=np.float32)
    rc = lib().orc_downsample(_p(conv), H, W, ch, rh, rw, _p(out))
    if rc != 0:
        raise ValueError("oracle: bad down-sampling geometry")
    return out


def to_u8(plane):
    plane = np.ascontiguousarray(plane, dtype=np.float32)
    out = np.empty(plane.shape, dtype=np.uint8)
    lib().orc_to_u8(_p(plane), _p(out), ctypes.c_int64(plane.size))
    return out


def _u8_stage(fn, img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty_like(img)
    getattr(lib(), fn)(_p(img), _p(out), img.shape[0], img.shape[1])
    return out


def clahe(img):
    return _u8_stage("orc_clahe", img)


def gauss3(img):
    return _u8_stage("orc_gauss3", img)


def bilateral5(img):
    return _u8_stage("orc_bilateral5", img)


def bilateral_tables():
    sw = np.empty(13, np.float32)
    dy = np.empty(13, np.int32)
    dx = np.empty(13, np.int32)
    cw = np.empty(256, np.float32)
    lib().orc_bilateral_tables(_p(sw), _p(dy), _p(dx), _p(cw))
    return sw, dy, dx, cw


def percentiles(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    lo, hi = ctypes.c_double(), ctypes.c_double()
    lib().orc_percentiles(_p(img), ctypes.c_int64(img.size), ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value


def canny_thresholds(lo, hi):
    a, b = ctypes.c_int(), ctypes.c_int()
    lib().orc_canny_thresholds(ctypes.c_double(lo), ctypes.c_double(hi), ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def canny(img, lo, hi, return_nms=False):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty_like(img)
    nms = np.empty_like(img) if return_nms else None
    lib().orc_canny(_p(img), _p(out), _p(nms) if return_nms else None, img.shape[0], img.shape[1],
                    ctypes.c_double(lo), ctypes.c_double(hi))
    return (out, nms) if return_nms else out


def edge_pipeline(plane, return_stages=False, params=None):
    """EdgeDetection.canny (edge_detection.py:70-86) -> uint8 {0,1}; optionally the 4 u8 stages + thresholds.
    params = (canny_low_ratio, canny_high_ratio, clahe_clip_limit, bilateral_sigma_color, bilateral_sigma_space, use_L2_gradient)
    of edge_detection.py:31-40, None = the defaults."""
    plane = np.ascontiguousarray(plane, dtype=np.float32)
    H, W = plane.shape
    edge = np.empty((H, W), np.uint8)
    stages = np.empty((4, H, W), np.uint8) if return_stages else None
    thr = (ctypes.c_double * 2)()
    if params is not None:
        pv = (ctypes.c_double * 6)(*[float(v) for v in params])
        lib().orc_edge_pipeline_ex(_p(plane), _p(edge), H, W, _p(stages) if return_stages else None, thr, pv)
    else:
        lib().orc_edge_pipeline(_p(plane), _p(edge), H, W, _p(stages) if return_stages else None, thr)
    if return_stages:
        return edge, stages, (thr[0], thr[1])
    return edge


def root_size(H, W):
    return lib().orc_root_size(H, W)


def quadtree(edge, min_size, max_size):
    """-> leaves (n,3) int32 [x,y,s], states (m,) uint8 {0 leaf,1 internal,2 absent}, root_size."""
    edge = np.ascontiguousarray(edge != 0, dtype=np.uint8)
    H, W = edge.shape
    root = root_size(H, W)
    cells = max(1, root // max(1, min(min_size, root)))
    cap = 2 * cells * cells + 64
    leaves = np.empty((cap, 3), np.int32)
    states = np.empty(2 * cap, np.uint8)
    nl, ns, r = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
    rc = lib().orc_quadtree(_p(edge), H, W, min_size, max_size, _p(leaves), ctypes.c_int64(cap), _p(states),
                            ctypes.c_int64(2 * cap), ctypes.byref(nl), ctypes.byref(ns), ctypes.byref(r))
    if rc != 0:
        raise RuntimeError("oracle quadtree capacity")
    return leaves[:nl.value].copy(), states[:ns.value].copy(), r.value


def normalize(plane, space, layer):
    mid = np.array(NORM[space][0], dtype=np.float32)[layer]
    sc = np.array(NORM[space][1], dtype=np.float32)[layer]
    plane = np.ascontiguousarray(plane, dtype=np.float32)
    out = np.empty_like(plane)
    lib().orc_normalize(_p(plane), _p(out), ctypes.c_int64(plane.size), ctypes.c_float(mid), ctypes.c_float(sc))
    return out


def dct_matrix(s):
    D = np.empty((s, s), np.float32)
    lib().orc_dct_matrix(s, _p(D))
    return D


def blocks_encode(norm, leaves, qm_by_size, zz_by_size, want_dct=False):
    """gather+reflect pad, DCT, quantise, zigzag for every leaf in order -> int32 coefficients (and raw DCT)."""
    norm = np.ascontiguousarray(norm, dtype=np.float32)
    leaves = np.ascontiguousarray(leaves, dtype=np.int32).reshape(-1, 3)
    H, W = norm.shape
    total = int((leaves[:, 2].astype(np.int64) ** 2).sum())
    coeffs = np.empty(total, np.int32)
    dct = np.empty(total, np.float32) if want_dct else None
    qarr = (ctypes.c_void_p * 16)()
    zarr = (ctypes.c_void_p * 16)()
    keep = []
    for s, q in qm_by_size.items():
        q = np.ascontiguousarray(q, dtype=np.int32)
        z = np.ascontiguousarray(zz_by_size[s], dtype=np.int32)
        keep += [q, z]
        lg = int(math.log2(s))
        qarr[lg] = q.ctypes.data
        zarr[lg] = z.ctypes.data
    rc = lib().orc_blocks_encode(_p(norm), H, W, _p(leaves), ctypes.c_int64(leaves.shape[0]), qarr, zarr,
                                 _p(coeffs), _p(dct) if want_dct else None)
    if rc != 0:
        raise RuntimeError(f"oracle blocks_encode rc={rc}")
    return (coeffs, dct) if want_dct else coeffs


# ------------------------------------------------------------------ host tables (a-14, a-15)
def zigzag(size):
    """Jpeg._zigzag_ordering (jpeg.py:726-766) restated as an anti-diagonal walk."""
    out = np.empty(size * size, np.int32)
    i = 0
    for d in range(2 * size - 1):
        lo, hi = max(0, d - size + 1), min(d, size - 1)
        rows = range(lo, hi + 1) if d % 2 else range(hi, lo - 1, -1)
        for r in rows:
            out[i] = r * size + (d - r)
            i += 1
    return out


def block_sizes(brange):
    return [2 ** i for i in range(int(math.log2(brange[0])), int(math.log2(brange[1])) + 1)]  # jpeg.py:219


def quality_factor(size, brange, qrange):
    """jpeg.py:688-705"""
    bmin, bmax = brange
    qmin, qmax = qrange
    if bmin == bmax:
        return int((qmin + qmax) / 2)
    return int(qmin + (qmax - qmin) * (1 - math.log(size / bmin) / math.log(bmax / bmin)))


def resize_linear_f32(src, size):
    """cv.resize(src f32 8x8, (size,size), INTER_LINEAR) as OpenCV 4.x computes it (resize.cpp):
    exact 2x shrink -> 2x2 box mean (INTER_AREA fast path); otherwise half-pixel-centre bilinear with
    the source index clamped at both ends, horizontal pass then vertical pass in float32."""
    src = np.asarray(src, dtype=np.float32)
    n = src.shape[0]
    if size == n:
        return src.copy()
    if n == 2 * size:
        return ((src[0::2, 0::2] + src[0::2, 1::2]) + (src[1::2, 0::2] + src[1::2, 1::2])) * np.float32(0.25)
    scale = n / size
    idx = np.empty(size, np.int64)
    fr = np.empty(size, np.float32)
    for d in range(size):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= n - 1:
            s, f = n - 1, np.float32(0)
        idx[d], fr[d] = s, f
    idx1 = np.minimum(idx + 1, n - 1)
    a0 = (np.float32(1) - fr).astype(np.float32)
    hor = src[:, idx] * a0[None, :] + src[:, idx1] * fr[None, :]
    out = hor[idx, :] * a0[:, None] + hor[idx1, :] * fr[:, None]
    return out.astype(np.float32)


def quant_matrix(table, size, quality):
    """Jpeg._get_quantization_matrix (jpeg.py:707-724)"""
    scale_factor = 5000 / quality if quality < 50 else 200 - 2 * quality
    scaled = np.floor((scale_factor * table + 50) / 100)
    resized = resize_linear_f32(scaled, size)
    return np.clip(resized, 1, None).astype(np.int32)


def layer_shapes(H, W, space):
    return [(H // rh, W // rw) for rh, rw in RATIOS[space]]  # jpeg.py:676-686


def tables(space, qrange, brange):
    sizes = block_sizes(brange)
    zz = {s: zigzag(s) for s in sizes}
    qm = []
    for layer in range(3):
        t = LUM if layer == 0 else CHR
        qm.append({s: quant_matrix(t, s, quality_factor(s, brange, qrange)) for s in sizes})
    return sizes, zz, qm


# ------------------------------------------------------------------ whole path (jpeg.py:240-272)
def encode_image(rgb, space="YCoCg", qrange=(40, 80), brange=(4, 64), keep=False):
    """rgb: (H,W,3) float32 in [0,1].  Returns per-layer dicts: root_size, states, leaves, coeffs."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    H, W, _ = rgb.shape
    conv = color_forward(space, rgb.reshape(-1, 3)).reshape(H, W, 3)
    _, zz, qm = tables(space, qrange, brange)
    layers = []
    for i, (rh, rw) in enumerate(RATIOS[space]):
        plane = downsample(conv, i, rh, rw)
        edge = edge_pipeline(plane)
        leaves, states, root = quadtree(edge, brange[0], brange[1])
        norm = normalize(plane, space, i)
        coeffs = blocks_encode(norm, leaves, qm[i], zz)
        d = {"root_size": root, "states": states, "leaves": leaves, "coeffs": coeffs}
        if keep:
            d.update(plane=plane, edge=edge, norm=norm)
        layers.append(d)
    return layers


def write_ajpg(layers, H, W, space, qrange, brange, extension):
    """Jpeg._entropy_encode (jpeg.py:531-597) -- next-scope bitstream writer, restated for fixtures."""
    out = BytesIO()
    meta = {"height": H, "width": W, "num_layers": len(layers), "color_space": space,
            "quality_min": qrange[0], "quality_max": qrange[1],
            "block_size_min": brange[0], "block_size_max": brange[1], "extension": extension}
    mb = json.dumps(meta).encode("utf-8")
    out.write(len(mb).to_bytes(4, "big"))
    out.write(mb)
    for L in layers:
        st = np.asarray(L["states"], dtype=np.uint8)
        bits_len = 2 * len(st)
        pad = (-len(st)) % 4
        sp = np.concatenate([st, np.zeros(pad, np.uint8)]).reshape(-1, 4)
        packed = ((sp[:, 0] << 6) | (sp[:, 1] << 4) | (sp[:, 2] << 2) | sp[:, 3]).astype(np.uint8)
        out.write(bits_len.to_bytes(4, "big"))
        out.write(int(L["root_size"]).to_bytes(4, "big"))
        out.write(packed.tobytes())
        comp = zlib.compress(np.ascontiguousarray(L["coeffs"], dtype=np.int32).tobytes(), level=9)
        out.write(len(comp).to_bytes(4, "big"))
        out.write(comp)
    return out.getvalue()


def synth_image(H, W, seed, kind="mixed"):
    """The synthetic benchmark image of SURVEY.md section 8d (uint8 HxWx3)."""
    rng = np.random.default_rng(seed)
    if kind == "flat":
        return np.full((H, W, 3), 128, np.uint8)
    if kind == "noise":
        return rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    yy = np.arange(H, dtype=np.float64)[:, None] / H
    xx = np.arange(W, dtype=np.float64)[None, :] / W
    img = np.empty((H, W, 3), np.float64)
    for c in range(3):
        fx, fy = rng.integers(1, 4, size=2)
        phi, psi = rng.uniform(0, 2 * np.pi, size=2)
        img[:, :, c] = 127.5 + 80.0 * np.sin(2 * np.pi * fx * xx + phi) * np.cos(2 * np.pi * fy * yy + psi)
    K = -(-(H * W) // 32768)
    for _ in range(K):
        x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
        w, h = int(rng.integers(16, 257)), int(rng.integers(16, 257))
        col = rng.integers(0, 256, size=3)
        img[y0:y0 + h, x0:x0 + w, :] = col
    img += rng.normal(0.0, 1.5, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ decode path (jpeg.py:274-297), next-scope row
def color_inverse(space, data):
    data = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, 3)
    out = np.empty_like(data)
    lib().orc_color_inverse(ctypes.c_int(SPACES[space]), _p(data), _p(out), ctypes.c_int64(data.shape[0]))
    return out


def decode_leaf_sizes(states, root):
    """Jpeg._decode_leaf_sizes (jpeg.py:768-800)"""
    sizes, stack, i = [], [root], 0
    states = list(states)
    while stack and i < len(states):
        size = stack.pop()
        s = states[i]
        i += 1
        if s == 0:
            sizes.append(size)
        elif s == 1:
            stack.extend([size // 2] * 4)
    return sizes


def leaf_positions(sizes, root, H, W):
    sizes = np.ascontiguousarray(sizes, dtype=np.int32)
    xy = np.zeros((len(sizes), 2), np.int32)
    n = lib().orc_leaf_positions(_p(sizes), ctypes.c_int64(len(sizes)), root, H, W, _p(xy))
    assert n == len(sizes)
    return xy


def blocks_decode(coeffs, leaves, qm_by_size, zz_by_size, space, layer, H, W):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int32)
    leaves = np.ascontiguousarray(leaves, dtype=np.int32).reshape(-1, 3)
    plane = np.zeros((H, W), np.float32)
    qarr = (ctypes.c_void_p * 16)()
    zarr = (ctypes.c_void_p * 16)()
    keep = []
    for s, q in qm_by_size.items():
        q = np.ascontiguousarray(q, dtype=np.int32)
        z = np.ascontiguousarray(zz_by_size[s], dtype=np.int32)
        keep += [q, z]
        lg = int(math.log2(s))
        qarr[lg] = q.ctypes.data
        zarr[lg] = z.ctypes.data
    mid = np.array(NORM[space][0], dtype=np.float32)[layer]
    sc = np.array(NORM[space][1], dtype=np.float32)[layer]
    rc = lib().orc_blocks_decode(_p(coeffs), _p(leaves), ctypes.c_int64(leaves.shape[0]), qarr, zarr, ctypes.c_float(mid),
                                 ctypes.c_float(sc), H, W, _p(plane))
    if rc != 0:
        raise RuntimeError(f"oracle blocks_decode rc={rc}")
    return plane


def upsample_linear(plane, H, W):
    plane = np.ascontiguousarray(plane, dtype=np.float32)
    out = np.empty((H, W), np.float32)
    lib().orc_upsample_linear(_p(plane), plane.shape[0], plane.shape[1], _p(out), H, W)
    return out


def parse_ajpg(data):
    """Jpeg._entropy_decode's container parsing (jpeg.py:609-661) -> metadata, per-layer (states, root, coeffs)"""
    s = BytesIO(data)
    mlen = int.from_bytes(s.read(4), "big")
    meta = json.loads(s.read(mlen).decode("utf-8"))
    layers = []
    for _ in range(meta["num_layers"]):
        bits_len = int.from_bytes(s.read(4), "big")
        root = int.from_bytes(s.read(4), "big")
        packed = np.frombuffer(s.read((bits_len + 7) // 8), dtype=np.uint8)
        st = np.stack([(packed >> 6) & 3, (packed >> 4) & 3, (packed >> 2) & 3, packed & 3], 1).reshape(-1)[: bits_len // 2]
        clen = int.from_bytes(s.read(4), "big")
        coeffs = np.frombuffer(zlib.decompress(s.read(clen)), dtype=np.int32)
        layers.append({"states": st.astype(np.uint8), "root_size": root, "coeffs": coeffs})
    return meta, layers


def decode_image(data):
    """Jpeg.decompress (jpeg.py:274-297) -> float32 (H, W, 3) in [0, 1]"""
    meta, layers = parse_ajpg(data)
    H, W, space = meta["height"], meta["width"], meta["color_space"]
    qrange, brange = (meta["quality_min"], meta["quality_max"]), (meta["block_size_min"], meta["block_size_max"])
    _, zz, qm = tables(space, qrange, brange)
    planes = []
    for i, (L, (h, w)) in enumerate(zip(layers, layer_shapes(H, W, space))):
        sizes = decode_leaf_sizes(L["states"].tolist(), L["root_size"])
        root = root_size(h, w)                     # jpeg.py:425 recomputes it from the layer shape
        xy = leaf_positions(sizes, root, h, w)
        leaves = np.concatenate([xy, np.asarray(sizes, np.int32)[:, None]], 1)
        plane = blocks_decode(L["coeffs"], leaves, qm[i], zz, space, i, h, w)
        planes.append(upsample_linear(plane, H, W))
    img = np.stack(planes, axis=2)
    return color_inverse(space, img.reshape(-1, 3)).reshape(H, W, 3)
