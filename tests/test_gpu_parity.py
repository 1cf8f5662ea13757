"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs --
bit-exact for every integer / index output and for the float32 stages whose arithmetic order is defined.
Full-size cases (1080p, 4K) are checked through size-independent properties."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import adaptive_edge_aware_jpeg_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def ctx(A):
    from adaptive_edge_aware_jpeg_amd._lib import get_context
    return get_context()


def synth(oracle, H, W, seed, kind="mixed"):
    return oracle.synth_image(H, W, seed, kind).astype(np.float32) / np.float32(255.0)


# ------------------------------------------------------------------ a-1 colour
@pytest.mark.parametrize("space", ["YCbCr", "YCoCg", "YCoCg-R", "OKLAB", "ICtCp", "ICaCb", "JzAzBz"])
def test_colour_convert_bit_exact(A, oracle, space):
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, size=(50000, 3)).astype(np.float32) / np.float32(255.0)
    x[:256] = (np.arange(256, dtype=np.float32) / np.float32(255.0))[:, None]        # greys: exact-integer luma
    assert np.array_equal(A.convert("sRGB", space, x), oracle.color_forward(space, x))


def test_oklab_against_a_path_that_shares_no_code_with_the_kernel(A, oracle):
    """the oracle's OKLAB cube root is the kernel's own operation sequence (bit-exact test above); here the GPU output is
    compared with the oracle running its general float64 pow instead -- a checker with nothing in common with csrc/aej_devmath.h."""
    rng = np.random.default_rng(8)
    x = rng.integers(0, 256, size=(100000, 3)).astype(np.float32) / np.float32(255.0)
    oracle.set_oklab_independent_pow(True)
    try:
        ref = oracle.color_forward("OKLAB", x)
    finally:
        oracle.set_oklab_independent_pow(False)
    got = A.convert("sRGB", "OKLAB", x)
    assert np.abs(got - ref).max() <= 3e-7 and (got != ref).mean() < 0.02


def test_colour_against_reference_golden(A):
    c = np.load(os.path.join(GOLDEN, "color_forward.npz"))
    x = c["rgb_u8"].astype(np.float32) / np.float32(255.0)
    for sp in ("YCbCr", "YCoCg", "YCoCg-R", "ICaCb"):
        assert np.array_equal(A.convert("sRGB", sp, x), c[sp])
    for sp, tol in (("OKLAB", 6e-7), ("ICtCp", 1e-9), ("JzAzBz", 1e-9)):
        assert np.abs(A.convert("sRGB", sp, x) - c[sp]).max() <= tol


def test_colour_round_trip_closure_property(A, oracle):
    """reference test_color_conversions.py:67-68 asserts round-trip error < 1e-4; the inverse transforms are decode
    scope, so close the loop with float64 matrix inverses for the linear spaces."""
    rng = np.random.default_rng(4)
    x = rng.random((4096, 3), dtype=np.float32)
    M = {"YCbCr": [[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5], [0.5, -0.418688, -0.081312]],
         "YCoCg": [[0.25, 0.5, 0.25], [0.5, 0, -0.5], [-0.25, 0.5, -0.25]]}
    for sp, m in M.items():
        y = A.convert("sRGB", sp, x).astype(np.float64)
        back = y @ np.linalg.inv(np.array(m)).T
        assert np.abs(back - x).max() < 1e-4


# ------------------------------------------------------------------ a-2/a-3/a-11 fused plane kernel
@pytest.mark.parametrize("space,H,W,levels", [("YCbCr", 64, 128, "u8"), ("YCoCg", 130, 260, "u8"), ("ICtCp", 34, 72, "u8"), ("OKLAB", 18, 36, "u8"),
                                                ("JzAzBz", 16, 20, "u8"), ("OKLAB", 64, 256, "float"), ("ICaCb", 32, 128, "mixed"),
                                                ("JzAzBz", 48, 132, "mixed")])
def test_colour_planes_kernel(A, ctx, oracle, space, H, W, levels):
    """levels: 'u8' = every channel is k / 255 (the tabulated sRGB linearisation serves the whole image), 'float' = arbitrary
    float32 values in [0, 1) (every lane takes the float64 pow), 'mixed' = both within the same waves."""
    import torch
    img = synth(oracle, H, W, 5)
    if levels != "u8":
        rnd = np.random.default_rng(3).random((H, W, 3), dtype=np.float32)
        if levels == "float":
            img = rnd
        else:
            pick = np.random.default_rng(4).random((H, W, 1)) < 0.3
            img = np.where(pick, rnd, img).astype(np.float32)
    j = A.Jpeg(A.JpegCompressionSettings(space))
    c = j._bind()
    plan = c.plan(1, H, W)
    n = sum(((plan.layer_h[l] * plan.layer_w[l] + 63) // 64) * 64 for l in range(3))
    raw, nrm, u8 = c.empty((n,), torch.float32), c.empty((n,), torch.float32), c.empty((n,), torch.uint8)
    x = c.to_device(img[None], torch.float32)
    c.check(c.lib.aej_color_planes(c.handle, x.data_ptr(), 1, H, W, raw.data_ptr(), nrm.data_ptr(), u8.data_ptr()))
    raw, nrm, u8 = raw.cpu().numpy(), nrm.cpu().numpy(), u8.cpu().numpy()
    conv = oracle.color_forward(space, img.reshape(-1, 3)).reshape(H, W, 3)
    off = 0
    for l, (rh, rw) in enumerate(oracle.RATIOS[space]):
        pl = oracle.downsample(conv, l, rh, rw)
        m = pl.size
        assert (plan.layer_h[l], plan.layer_w[l]) == pl.shape
        assert np.array_equal(raw[off:off + m].reshape(pl.shape), pl)
        assert np.array_equal(nrm[off:off + m].reshape(pl.shape), oracle.normalize(pl, space, l))
        assert np.array_equal(u8[off:off + m].reshape(pl.shape), oracle.to_u8(pl))
        off += ((m + 63) // 64) * 64


# ------------------------------------------------------------------ a-3..a-8 Canny chain, stage by stage
@pytest.mark.parametrize("H,W,seed,shift", [(96, 160, 1, 0.0), (67, 101, 2, -0.5), (270, 480, 3, 0.0), (33, 50, 4, 0.0),
                                            (512, 512, 5, 0.0), (5, 7, 6, 0.0), (64, 64, 7, 0.0), (130, 70, 8, 0.3)])
def test_canny_chain_stages(A, ctx, oracle, H, W, seed, shift):
    img = synth(oracle, H, W, seed)
    plane = oracle.color_forward("YCbCr", img.reshape(-1, 3)).reshape(H, W, 3)[:, :, 0].copy() + np.float32(shift)
    e, st, thr = A.EdgeDetection.canny(plane, return_stages=True)
    eo, so, pct = oracle.edge_pipeline(plane, return_stages=True)
    for i, name in enumerate(("scaled", "clahe", "gauss", "bilateral")):
        assert np.array_equal(st[i], so[i]), name
    assert thr == oracle.canny_thresholds(*pct)
    _, nms = oracle.canny(so[3], pct[0], pct[1], return_nms=True)
    assert np.array_equal(st[4], nms), "nms map"
    assert e.dtype == np.float32 and np.array_equal(e.astype(np.uint8), eo)


@pytest.mark.parametrize("params", [
    dict(canny_low_ratio=0.20, canny_high_ratio=0.55),
    dict(clahe_clip_limit=2.0), dict(clahe_clip_limit=40.0), dict(clahe_clip_limit=0.0),
    dict(bilateral_sigma_color=20, bilateral_sigma_space=3.5),
    dict(use_L2_gradient=False),
    dict(use_L2_gradient=False, canny_low_ratio=0.6, canny_high_ratio=0.4, clahe_clip_limit=1.5, bilateral_sigma_color=150, bilateral_sigma_space=0),
])
def test_canny_hyper_parameters(A, oracle, params):
    """EdgeDetection.canny's keyword arguments (edge_detection.py:31-40): the run-time ones against the oracle stage by stage,
    the structural ones raise; a call with other values leaves the shared context on the defaults."""
    H, W = 203, 333
    img = synth(oracle, H, W, 11)
    plane = oracle.color_forward("YCbCr", img.reshape(-1, 3)).reshape(H, W, 3)[:, :, 0].copy()
    full = dict(canny_low_ratio=0.10, canny_high_ratio=0.30, clahe_clip_limit=0.75, bilateral_sigma_color=75, bilateral_sigma_space=75, use_L2_gradient=True)
    full.update(params)
    tup = (full["canny_low_ratio"], full["canny_high_ratio"], full["clahe_clip_limit"], full["bilateral_sigma_color"], full["bilateral_sigma_space"],
           full["use_L2_gradient"])
    e, st, thr = A.EdgeDetection.canny(plane, return_stages=True, **params)
    eo, so, pct = oracle.edge_pipeline(plane, return_stages=True, params=tup)
    for i, name in enumerate(("scaled", "clahe", "gauss", "bilateral")):
        assert np.array_equal(st[i], so[i]), name
    assert np.array_equal(e.astype(np.uint8), eo)
    assert eo.sum() > 0
    # defaults restored
    assert np.array_equal(A.EdgeDetection.canny(plane).astype(np.uint8), oracle.edge_pipeline(plane))
    for bad in (dict(aperture_size=5), dict(clahe_tile_grid=(8, 8)), dict(bilateral_diameter=9), dict(gaussian_kernel=5)):
        with pytest.raises(NotImplementedError):
            A.EdgeDetection.canny(plane, **bad)


def test_canny_extremes(A, oracle):
    flat = np.full((128, 192), 0.5, np.float32)
    assert A.EdgeDetection.canny(flat).sum() == 0
    noise = np.random.default_rng(1).random((128, 192), dtype=np.float32)
    assert np.array_equal(A.EdgeDetection.canny(noise).astype(np.uint8), oracle.edge_pipeline(noise))


def test_hysteresis_long_chain(A, oracle):
    """a weak spiral many tiles long with a single strong seed: needs many hysteresis passes."""
    H = W = 400
    plane = np.full((H, W), 0.30, np.float32)
    y = x = 10
    dx, dy, run = 1, 0, 380
    while run > 8:                                   # rectangular spiral of a faint line
        for _ in range(run):
            plane[y, x] += 0.035
            x += dx
            y += dy
        dx, dy = -dy, dx
        run -= 12
    plane[10, 10:14] = 0.9                            # the strong seed
    got = A.EdgeDetection.canny(plane).astype(np.uint8)
    assert np.array_equal(got, oracle.edge_pipeline(plane))


@pytest.mark.parametrize("kind", ["noise", "maze", "texture"])
def test_hysteresis_many_tiles_concurrent_chases(A, oracle, kind):
    """Passes >= 1 let a wave follow a contour through up to 32 tiles while other waves work on the same tiles (atomicOr
    writes); the fix-point must still be OpenCV's flood fill.  Large planes whose final edges are mostly weak pixels promoted
    through long chains (maze: 1.8 K strong seeds -> 155 K edge pixels along 180 serpentine lines of slowly rising contrast)."""
    H, W = 1100, 1500
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    if kind == "noise":
        plane = rng.random((H, W), dtype=np.float32)
    elif kind == "texture":
        plane = (0.12 + 0.1 * np.sin(xx / 11.0 + yy / 17.0) * np.cos(yy / 7.0)).astype(np.float32)
    else:
        plane = np.repeat(np.linspace(0.0, 0.2, H, dtype=np.float32)[:, None], W, axis=1).copy()
        amp = 0.036 + (0.05 - 0.036) * np.arange(W, dtype=np.float32) / W
        for k, y in enumerate(range(8, H - 8, 6)):
            plane[y, 6:W - 6] += amp[6:W - 6]
            x = W - 7 if k % 2 == 0 else 6
            plane[y:y + 6, x] += amp[x]
        plane = plane.astype(np.float32)
    edge, stages, thr = oracle.edge_pipeline(plane, return_stages=True)
    _, nms = oracle.canny(stages[3], thr[0], thr[1], return_nms=True)
    strong, weak = int((nms == 2).sum()), int((nms == 0).sum())
    assert weak > 50000 and int(edge.sum()) > strong + 40000, "the pattern must depend on hysteresis propagation"
    got = A.EdgeDetection.canny(plane).astype(np.uint8)
    assert np.array_equal(got, edge)


def test_whole_path_hysteresis_completes_on_the_device(A, ctx, oracle):
    """aej_encode_batch's hysteresis is a pass over every tile plus a device-side work queue drained by one launch: nothing is guessed
    or read back by the host.  An image whose edges are mostly weak pixels promoted through long chains across many tiles (the spiral of
    test_hysteresis_long_chain in every channel) must come out as the oracle's (the queue statistics are checked on the bench's images, test_gpu_full_size.py)."""
    H = W = 448
    plane = np.full((H, W), 0.30, np.float32)
    y = x = 10
    dx, dy, run = 1, 0, 420
    while run > 8:
        for _ in range(run):
            plane[y, x] += 0.035
            x += dx
            y += dy
        dx, dy = -dy, dx
        run -= 12
    plane[10, 10:14] = 0.9
    img = np.ascontiguousarray(np.repeat(plane[:, :, None], 3, axis=2))
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    ref = oracle.encode_image(img, "YCbCr", (40, 80), (4, 64))
    for _ in range(2):
        enc = codec.compress_batch(np.stack([img, img[::-1].copy()]))
        for l in range(3):
            got = enc.layer(0, l)
            assert np.array_equal(got["states"], ref[l]["states"]) and np.array_equal(got["leaves"], ref[l]["leaves"])
            assert np.array_equal(got["coeffs"], ref[l]["coeffs"])
    assert ctx.hysteresis_stats()["calls"] >= 2


def test_hysteresis_serpentine_across_one_tile_border(A, ctx, oracle):
    """ADVICE r4 asked for a contour that re-dirties the same few tiles again and again: a serpentine that crosses the border between two
    tile columns 72 times from a handful of seeds.  The fix-point must be the oracle's on the stand-alone entry and on the whole path.
    (It does NOT lap the work queue's ring: a wave follows a contour through 32 tiles before it queues anything, so this pattern costs 16
    queue entries on 22 tiles, a 448 x 448 maze 92 on 81, noise 18 on 24 -- the ring holds more than `tiles` entries, and no input we know
    pushes more than about one entry per tile.  What a lapped ring could do -- ADVICE r4's interlock -- is bounded instead: every wait of
    the queue gives up after about a second and the call returns AEJ_ERR_STATE, canny.hip kQPoison.)"""
    H, W = 448, 128                                   # 7 x 2 luma tiles (+ 4 x 1 per chroma layer): a ring of 32 slots
    plane = np.repeat(np.linspace(0.0, 0.2, H, dtype=np.float32)[:, None], W, axis=1).copy()
    amp = 0.036 + (0.05 - 0.036) * np.arange(W, dtype=np.float32) / W
    rows = list(range(8, H - 8, 6))
    for k, y in enumerate(rows):                      # faint horizontal lines across the tile border x = 64, joined at alternating ends
        plane[y, 6:W - 6] += amp[6:W - 6]
        x = W - 7 if k % 2 == 0 else 6
        plane[y:y + 6, x] += amp[x]
    edge, stages, thr = oracle.edge_pipeline(plane, return_stages=True)
    _, nms = oracle.canny(stages[3], thr[0], thr[1], return_nms=True)
    assert int(edge.sum()) > int((nms == 2).sum()) + 3000, "the pattern must depend on hysteresis propagation"
    assert np.array_equal(A.EdgeDetection.canny(plane).astype(np.uint8), edge)
    img = np.ascontiguousarray(np.repeat(plane[:, :, None], 3, axis=2))
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    ref = oracle.encode_image(img, "YCbCr", (40, 80), (4, 64))
    enc = codec.compress_batch(img[None])
    for l in range(3):
        got = enc.layer(0, l)
        assert np.array_equal(got["states"], ref[l]["states"]) and np.array_equal(got["leaves"], ref[l]["leaves"]) and np.array_equal(got["coeffs"], ref[l]["coeffs"])
    assert 0 < ctx.hysteresis_stats()["queued"] < (1 << 30)          # (bit 30 = a wait of the queue gave up: would have raised above)


# ------------------------------------------------------------------ a-9/a-10 quadtree
def test_quadtree_reference_golden_cases(A):
    g = np.load(os.path.join(GOLDEN, "quadtree_cases.npz"))
    n, checked = int(g["n_cases"][0]), 0
    for i in range(n):
        edge = g[f"c{i}_edge"].astype(np.float32)
        mn, mx, root = (int(v) for v in g[f"c{i}_params"])
        qt = A.QuadTree(edge, max_size=mx, min_size=mn)
        leaves, states = qt.get_leaves_and_states()
        assert qt.root_size == root
        assert [[l.x, l.y, l.size] for l in leaves] == g[f"c{i}_leaves"].tolist(), i
        assert [int(s, 2) for s in states] == g[f"c{i}_states"].tolist(), i
        checked += 1
    assert checked == n


def test_quadtree_tree_rebuild(A, oracle):
    rng = np.random.default_rng(2)
    edge = (rng.random((90, 130)) < 0.01).astype(np.float32)
    qt = A.QuadTree(edge, 32, 4)
    leaves, states = qt.get_leaves_and_states()
    found = []
    stack = [qt.root]
    while stack:                                      # quadtree.py:149-163
        node = stack.pop()
        if node is None:
            continue
        if node.is_leaf():
            found.append((node.x, node.y, node.size))
        else:
            stack.extend(reversed(node.children))
    assert found == [(l.x, l.y, l.size) for l in leaves]


@pytest.mark.parametrize("H,W,mn,mx", [(1080, 1920, 4, 64), (540, 960, 4, 64), (700, 1100, 8, 128), (333, 517, 2, 32)])
def test_quadtree_large_random(A, oracle, H, W, mn, mx):
    rng = np.random.default_rng(H)
    edge = (rng.random((H, W)) < 0.002).astype(np.float32)
    edge[H // 3, : W // 2] = 1.0
    qt = A.QuadTree(edge, max_size=mx, min_size=mn)
    lo, so, ro = oracle.quadtree(edge, mn, mx)
    assert np.array_equal(qt._leaves[:, :3], lo) and np.array_equal(qt._states, so) and qt.root_size == ro


# ------------------------------------------------------------------ a-11..a-15 DCT / quantise / zigzag
@pytest.mark.parametrize("bmin,bmax,H,W", [(4, 64, 150, 210), (4, 128, 260, 300), (8, 8, 64, 64), (2, 16, 40, 56), (32, 32, 70, 100)])
def test_dct_quant_zigzag(A, ctx, oracle, bmin, bmax, H, W):
    import torch
    rng = np.random.default_rng(9)
    space, qr = "YCbCr", (40, 80)
    j = A.Jpeg(A.JpegCompressionSettings(space, qr, (bmin, bmax)))
    c = j._bind()
    norm = (rng.random((H, W), dtype=np.float32) * 254 - 127).astype(np.float32)
    edge = (rng.random((H, W)) < 0.004).astype(np.uint8)
    leaves, _, _ = oracle.quadtree(edge, bmin, bmax)
    _, zz, qm = oracle.tables(space, qr, (bmin, bmax))
    offs = np.concatenate([[0], np.cumsum(leaves[:, 2].astype(np.int64) ** 2)[:-1]]).astype(np.int32)
    lv4 = np.concatenate([leaves, offs[:, None]], 1).astype(np.int32)
    for layer in (0, 2):
        co, do = oracle.blocks_encode(norm, leaves, qm[layer], zz, want_dct=True)
        d_l, d_n = c.to_device(lv4, torch.int32), c.to_device(norm, torch.float32)
        d_c, d_d = c.empty((co.size,), torch.int32), c.empty((co.size,), torch.float32)
        c.check(c.lib.aej_dct_quant_zigzag(c.handle, d_n.data_ptr(), H, W, layer, d_l.data_ptr(), ctypes.c_int64(len(lv4)),
                                           d_c.data_ptr(), d_d.data_ptr()))
        got_d, got_c = d_d.cpu().numpy(), d_c.cpu().numpy()
        # pre-quantisation DCT: contract is bit equality with the k-ordered fma chain (MFMA f32 == fmaf chain);
        # stated float tolerance vs the float64 DCT is 1e-6 * s * 127 (tests/test_oracle_pins.py)
        assert np.array_equal(got_d, do), f"layer {layer}: DCT floats"
        assert np.array_equal(got_c, co), f"layer {layer}: coefficients"


# ------------------------------------------------------------------ whole path
CASES = [("lena YCbCr 8-8 q50", "lena", "YCbCr", (50, 50), (8, 8)),
         ("lena YCoCg default", "lena", "YCoCg", (40, 80), (4, 64)),
         ("lena YCoCg-R 4-32", "lena", "YCoCg-R", (75, 75), (4, 32)),
         ("synth 360x640 YCbCr", (360, 640, 20250718), "YCbCr", (40, 80), (4, 64)),
         ("synth 250x332 OKLAB 4-128", (250, 332, 11), "OKLAB", (40, 80), (4, 128)),
         ("synth 128x256 ICtCp", (128, 256, 12), "ICtCp", (20, 60), (4, 32)),
         ("synth 96x128 ICaCb", (96, 128, 13), "ICaCb", (40, 80), (4, 16)),
         ("synth 120x200 JzAzBz", (120, 200, 14), "JzAzBz", (40, 80), (8, 64)),
         ("synth 1080p YCbCr", (1080, 1920, 20250718), "YCbCr", (40, 80), (4, 64)),
         # the reference's own natural test images (test_images/, metrics_computation.py:307-324): textures, many hysteresis passes
         ("baboon YCbCr", "natural/baboon", "YCbCr", (40, 80), (4, 64)),
         ("baboon OKLAB 4-128", "natural/baboon", "OKLAB", (40, 80), (4, 128)),
         ("peppers YCoCg default", "natural/peppers", "YCoCg", (40, 80), (4, 64)),
         ("peppers ICtCp", "natural/peppers", "ICtCp", (40, 80), (4, 64)),
         ("house JzAzBz 2-32", "natural/house", "JzAzBz", (30, 90), (2, 32)),
         ("jelly_beans ICaCb", "natural/jelly_beans", "ICaCb", (40, 80), (4, 64)),
         ("bikes YCbCr 4-128", "natural/bikes", "YCbCr", (40, 80), (4, 128)),
         ("buildings YCoCg-R 8-64", "natural/buildings", "YCoCg-R", (10, 95), (8, 64))]


@pytest.mark.parametrize("name,src,space,qr,br", CASES, ids=[c[0] for c in CASES])
def test_encode_matches_oracle(A, oracle, lena, name, src, space, qr, br):
    from conftest import golden_image
    img = lena if src == "lena" else golden_image(src) if isinstance(src, str) else synth(oracle, *src)
    enc = A.Jpeg(A.JpegCompressionSettings(space, qr, br)).compress_batch(img[None], want_dct=True)
    ref = oracle.encode_image(img, space, qr, br)
    for l in range(3):
        got = enc.layer(0, l)
        assert got["root_size"] == ref[l]["root_size"]
        assert np.array_equal(got["states"], ref[l]["states"]), f"L{l} states"
        assert np.array_equal(got["leaves"], ref[l]["leaves"]), f"L{l} leaves"
        assert np.array_equal(got["coeffs"], ref[l]["coeffs"]), f"L{l} coeffs"


RAGGED = [("YCbCr", 33, 35), ("YCbCr", 101, 67), ("YCoCg", 151, 211), ("YCbCr", 66, 130), ("ICtCp", 37, 50), ("ICaCb", 40, 51),
          ("OKLAB", 21, 23), ("YCbCr", 5, 9), ("JzAzBz", 64, 62)]


@pytest.mark.parametrize("space,H,W", RAGGED, ids=[f"{c[0]}-{c[1]}x{c[2]}" for c in RAGGED])
def test_ragged_sizes_general_inter_area(A, ctx, oracle, space, H, W):
    """odd / non-multiple-of-4 sizes: cv.resize(INTER_AREA) becomes the general area-weighted resize, planes need
    CLAHE reflect padding, leaves overhang: whole path against the oracle."""
    img = synth(oracle, H, W, H * 1000 + W)
    br = (4, 32)
    enc = A.Jpeg(A.JpegCompressionSettings(space, (40, 80), br)).compress_batch(np.stack([img, img[::-1].copy()]))
    for b, im in enumerate((img, img[::-1].copy())):
        ref = oracle.encode_image(im, space, (40, 80), br)
        for l in range(3):
            got = enc.layer(b, l)
            assert got["root_size"] == ref[l]["root_size"]
            assert np.array_equal(got["states"], ref[l]["states"]), f"L{l} states"
            assert np.array_equal(got["leaves"], ref[l]["leaves"]), f"L{l} leaves"
            assert np.array_equal(got["coeffs"], ref[l]["coeffs"]), f"L{l} coeffs"


def test_compress_bytes_match_reference_orchestrated_fixture(A, lena):
    """Jpeg.compress(Image) -> .ajpg identical to the files the reference's own compress() wrote (with the oracle
    as its cv2)."""
    meta = json.load(open(os.path.join(GOLDEN, "compress_cases.json")))
    from conftest import golden_image
    for name, m in meta.items():
        img = lena if m["image"] == "lena" else golden_image(m["image"])
        if m["crop"]:
            y, x, h, w = m["crop"]
            img = np.ascontiguousarray(img[y:y + h, x:x + w])
        codec = A.Jpeg(A.JpegCompressionSettings(m["space"], tuple(m["quality_range"]), tuple(m["block_size_range"])))
        data = codec.compress(A.Image(img, img.shape, ".png"))
        assert hashlib.sha256(data).hexdigest() == m["sha256"], name
        assert A.Jpeg._decode_leaf_sizes  # header decoder of the reference stays available


def test_batch_equals_single_and_is_deterministic(A, oracle):
    imgs = np.stack([synth(oracle, 256, 384, s, k) for s, k in ((1, "mixed"), (2, "noise"), (3, "flat"), (4, "mixed"))])
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    batch = codec.compress_batch(imgs)
    again = codec.compress_batch(imgs)
    for b in range(len(imgs)):
        single = codec.compress_batch(imgs[b:b + 1])
        for l in range(3):
            x, y, z = batch.layer(b, l), single.layer(0, l), again.layer(b, l)
            for k in ("states", "leaves", "coeffs"):
                assert np.array_equal(x[k], y[k]) and np.array_equal(x[k], z[k])
    flat, noise = batch.layer(2, 0), batch.layer(1, 0)
    assert set(flat["leaves"][:, 2].tolist()) == {64}                 # constant image: all max-size leaves
    assert np.count_nonzero(flat["coeffs"]) <= len(flat["leaves"])    # only DC terms survive
    assert (noise["leaves"][:, 2] == 4).mean() > 0.5 and 64 not in set(noise["leaves"][:, 2].tolist())   # white noise: small leaves


def test_compress_many_equals_compress(A, oracle):
    """batch -> list of .ajpg byte strings (GPU pass + thread-pooled zlib-9) == Jpeg.compress image by image"""
    imgs = np.stack([synth(oracle, 200, 264, s, k) for s, k in ((1, "mixed"), (2, "noise"), (3, "flat"))])
    codec = A.Jpeg(A.JpegCompressionSettings("YCoCg", (40, 80), (4, 64)))
    many = codec.compress_many(imgs, extension=".png", workers=4)
    for i in range(len(imgs)):
        assert many[i] == codec.compress(A.Image(imgs[i], imgs[i].shape, ".png"))
    assert np.array_equal(A.Jpeg(A.JpegCompressionSettings()).decompress(many[0]).data, codec.decompress_batch(codec.compress_batch(imgs[:1])).cpu().numpy()[0])


@pytest.mark.parametrize("space,H,W", [("YCbCr", 256, 384), ("ICtCp", 128, 256), ("OKLAB", 67, 101), ("YCoCg", 50, 33)])
def test_uint8_ingest_equals_float_ingest(A, oracle, space, H, W):
    """aej_encode_batch_u8 forms float32(v) / 255 itself (image.py:80): identical to the float path and to the oracle,
    on the 4x2-patch kernel and on the general INTER_AREA kernel (ragged sizes)."""
    u8 = np.stack([oracle.synth_image(H, W, 11 + i, k) for i, k in enumerate(("mixed", "noise"))]).astype(np.uint8)
    f32 = u8.astype(np.float32) / np.float32(255.0)
    codec = A.Jpeg(A.JpegCompressionSettings(space, (40, 80), (4, 64)))
    a, b = codec.compress_batch(u8), codec.compress_batch(f32)
    for i in range(len(u8)):
        ref = oracle.encode_image(f32[i], space, (40, 80), (4, 64))
        for l in range(3):
            x, y = a.layer(i, l), b.layer(i, l)
            for k in ("states", "leaves", "coeffs"):
                assert np.array_equal(x[k], y[k]), (space, i, l, k)
            assert np.array_equal(x["coeffs"], ref[l]["coeffs"]) and np.array_equal(x["leaves"], ref[l]["leaves"])


@pytest.mark.parametrize("H,W,B", [(1080, 1920, 2), (2160, 3840, 1)])
def test_full_size_properties(A, oracle, H, W, B):
    """BASELINE sizes: structural invariants instead of an oracle run per pixel (the 1080p oracle comparison is
    in test_encode_matches_oracle)."""
    imgs = np.stack([synth(oracle, H, W, 20250718 + i) for i in range(B)])
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    enc = codec.compress_batch(imgs)
    for b in range(B):
        for l in range(3):
            L = enc.layer(b, l)
            h, w = (H, W) if l == 0 else (H // 2, W // 2)
            lv = L["leaves"]
            cover = np.zeros((h + 64, w + 64), np.int16)
            for s in np.unique(lv[:, 2]):
                sel = lv[lv[:, 2] == s]
                for dy in range(0, s, 4):               # paint in 4x4 cells, vectorised over leaves
                    for dx in range(0, s, 4):
                        np.add.at(cover, (sel[:, 1] + dy, sel[:, 0] + dx), 1)
            assert np.all(cover[:h:4, :w:4][: (h + 3) // 4, : (w + 3) // 4] == 1), "leaves must tile the plane exactly once"
            assert L["coeffs"].size == int((lv[:, 2].astype(np.int64) ** 2).sum())
            assert A.Jpeg._decode_leaf_sizes(L["states"].tolist(), L["root_size"]) == lv[:, 2].tolist()
            assert np.array_equal(L["leaf_coeff_offsets"], np.concatenate([[0], np.cumsum(lv[:, 2].astype(np.int64) ** 2)[:-1]]))
            # Morton (x in even bits) order of the leaf origins == DFS order
            def morton(x, y):
                m = np.zeros_like(x, dtype=np.int64)
                for bit in range(13):
                    m |= ((x >> bit) & 1).astype(np.int64) << (2 * bit)
                    m |= ((y >> bit) & 1).astype(np.int64) << (2 * bit + 1)
                return m
            mc = morton(lv[:, 0], lv[:, 1])
            assert np.all(np.diff(mc) > 0)


# ------------------------------------------------------------------ decode path (next-scope row)
@pytest.mark.parametrize("space", ["YCbCr", "YCoCg", "YCoCg-R", "OKLAB", "ICtCp", "ICaCb", "JzAzBz"])
def test_colour_inverse_bit_exact(A, oracle, space):
    g = np.load(os.path.join(GOLDEN, "color_inverse.npz"))
    x = g[space + "_in"]
    got = A.convert(space, "sRGB", x)
    assert np.array_equal(got, oracle.color_inverse(space, x))
    if space != "OKLAB":
        assert np.array_equal(got, g[space])                      # the reference's own output
    else:
        assert np.abs(got - g[space]).max() < 2e-5
    # reference test_color_conversions.py:67-68: round trip closes to 1e-4
    rgb = np.random.default_rng(1).random((4096, 3), dtype=np.float32)
    back = A.convert(space, "sRGB", A.convert("sRGB", space, rgb))
    assert np.abs(back - rgb).max() < 1e-4


def test_decompress_fixtures(A, oracle):
    meta = json.load(open(os.path.join(GOLDEN, "decode_cases.json")))
    for name, m in meta.items():
        data = open(os.path.join(GOLDEN, name + ".ajpg"), "rb").read()
        img = A.Jpeg(A.JpegCompressionSettings()).decompress(data)      # fresh default codec, settings come from the header
        assert isinstance(img, A.Image) and img.extension == m["extension"] and list(img.data.shape) == m["shape"]
        assert np.array_equal(img.data, oracle.decode_image(data))
        assert hashlib.sha256(np.ascontiguousarray(img.data).tobytes()).hexdigest() == m["sha256"], name


ROUNDTRIP = [("YCbCr", 256, 384, (4, 64)), ("OKLAB", 250, 332, (4, 128)), ("ICtCp", 128, 256, (4, 32)), ("ICaCb", 96, 128, (4, 16)),
             ("JzAzBz", 120, 200, (8, 64)), ("YCoCg", 101, 67, (4, 32)), ("YCoCg-R", 33, 35, (2, 16)), ("YCbCr", 1080, 1920, (4, 64))]


@pytest.mark.parametrize("space,H,W,br", ROUNDTRIP, ids=[f"{c[0]}-{c[1]}x{c[2]}" for c in ROUNDTRIP])
def test_device_round_trip_matches_oracle(A, oracle, space, H, W, br):
    img = synth(oracle, H, W, 77 + H)
    codec = A.Jpeg(A.JpegCompressionSettings(space, (40, 80), br))
    enc = codec.compress_batch(np.stack([img, img[:, ::-1].copy()]))
    dec = codec.decompress_batch(enc).cpu().numpy()
    for b, im in enumerate((img, img[:, ::-1].copy())):
        layers = oracle.encode_image(im, space, (40, 80), br)
        ref = oracle.decode_image(oracle.write_ajpg(layers, H, W, space, (40, 80), br, ".png"))
        assert np.array_equal(dec[b], ref)
        assert 10 * np.log10(1.0 / np.mean((dec[b] - im) ** 2)) > (25.0 if H * W > 10000 else 18.0)
    # bytes round trip through the reference-compatible container
    data = codec.compress(A.Image(img, img.shape, ".png"))
    assert np.array_equal(A.Jpeg(A.JpegCompressionSettings()).decompress(data).data, dec[0])


@pytest.mark.parametrize("space,H,W,br", [("YCbCr", 600, 800, (8, 256)), ("YCoCg", 530, 520, (4, 256)), ("OKLAB", 300, 700, (256, 256)),
                                         ("YCbCr", 1100, 1300, (8, 512)), ("YCoCg", 640, 1500, (512, 512)), ("YCbCr", 1200, 2300, (8, 1024))])
def test_block_size_256(A, oracle, space, H, W, br):
    """The GUI's largest block (main_frame.py block-size slider) and the larger powers of two the reference's Jpeg itself accepts
    (jpeg.py:216-219): 256 / 512 / 1024 leaves, whole and clipped at the plane border (np.pad reflect), through the tiled
    big-block kernels -- encode and decode against the oracle."""
    img = oracle.synth_image(H, W, 3, "flat").astype(np.float32) / np.float32(255.0)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = img + (0.05 * np.sin(xx / 90.0) * np.cos(yy / 70.0))[..., None].astype(np.float32)      # smooth, no edges
    img[100:140, 200:260] = 0.9                                                                    # one patch with edges
    img = np.clip(np.round(img * 255), 0, 255).astype(np.float32) / np.float32(255.0)
    codec = A.Jpeg(A.JpegCompressionSettings(space, (40, 80), br))
    enc = codec.compress_batch(img[None], want_dct=True)
    ref = oracle.encode_image(img, space, (40, 80), br)
    assert any(br[1] in set(ref[l]["leaves"][:, 2].tolist()) for l in range(3))
    for l in range(3):
        got = enc.layer(0, l)
        assert np.array_equal(got["states"], ref[l]["states"]) and np.array_equal(got["leaves"], ref[l]["leaves"])
        assert np.array_equal(got["coeffs"], ref[l]["coeffs"]), f"L{l} coeffs"
    dec = codec.decompress_batch(enc).cpu().numpy()[0]
    want = oracle.decode_image(oracle.write_ajpg(ref, H, W, space, (40, 80), br, ".png"))
    assert np.array_equal(dec, want)


def test_unsupported_inputs_fail_loudly(A):
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr"))
    with pytest.raises(NotImplementedError):
        A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (16, 2048))).compress_batch(np.zeros((1, 64, 64, 3), np.float32))   # no kernel above 1024
    with pytest.raises(NotImplementedError):
        A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (2, 512))).compress_batch(np.zeros((1, 64, 64, 3), np.float32))     # more than 8 sizes
    with pytest.raises(ValueError):
        codec.compress_batch(np.zeros((64, 64, 3), np.float32))


# ------------------------------------------------------------------ opt-in GPU entropy stage (csrc/deflate.hip)
def test_gpu_deflate_streams_are_zlib_streams_of_the_coefficients(A, oracle):
    """aej_deflate_histogram / aej_deflate_batch: every layer's stream must be a conforming zlib stream -- `zlib.decompress`, the call
    the reference's decoder makes (jpeg.py:659), returns exactly the layer's int32 coefficients -- for smooth, noisy, natural and ragged
    images, a batch, and sizes that leave partial 32 KiB chunks and partial sub-blocks; the container built from them decodes to the
    same image as the default (host zlib-9) container, with our decoder AND with the oracle's restatement of the reference's.  The
    matcher must really find far matches (distance histogram) and land near zlib's size on natural images."""
    import zlib
    import deflate_reference as DT
    from conftest import golden_image
    rng = np.random.default_rng(5)
    cases = [("YCbCr", (40, 80), (4, 64), np.stack([synth(oracle, 200, 328, 3), synth(oracle, 200, 328, 4)])),
             ("YCoCg", (5, 95), (2, 32), rng.random((1, 123, 77, 3), dtype=np.float32)),                      # noise: few zeros, many literals >= 144
             ("YCbCr", (40, 80), (4, 64), golden_image("natural/baboon")[None]),
             ("OKLAB", (40, 80), (8, 128), np.full((1, 256, 256, 3), 0.5, np.float32)),                          # flat: one long zero run per layer
             ("YCbCr", (90, 99), (4, 4), golden_image("natural/peppers")[None, :300, :420])]
    for space, qr, br, batch in cases:
        codec = A.Jpeg(A.JpegCompressionSettings(space, qr, br))
        enc = codec.compress_batch(batch)
        sizes = {}
        for adaptive in (False, True):
            streams = codec.deflate_batch(enc, adaptive=adaptive)
            again = codec.deflate_batch(enc, adaptive=adaptive)
            for b in range(batch.shape[0]):
                for l in range(3):
                    raw = enc.layer(b, l)["coeffs"].tobytes()
                    assert streams[b][l][:1] == b"\x78"
                    assert zlib.decompress(streams[b][l]) == raw, f"{space} image {b} layer {l}: {len(raw)} bytes, adaptive={adaptive}"
                    assert streams[b][l] == again[b][l], "the same input must give the same bytes"
            sizes[adaptive] = sum(len(x) for im in streams for x in im)
        # a stream takes the dynamic code only where it is smaller: never larger than all-fixed
        assert sizes[True] <= sizes[False], sizes
        # codes counted on OTHER data, without codes for symbols that data did not contain (a run of zero coefficients): a stream that
        # needs a missing code falls back to the fixed code -- still this batch's bytes
        zt = [("lit", 0)] + [("match", 258, 1)] * 15
        zh = DT.histogram_of(zt)
        foreign = np.stack([DT.adaptive_table(zh[:286], zh[286:316], cover_all=False)] * 3)
        streams_f = codec.deflate_batch(enc, tables=foreign)
        for b in range(batch.shape[0]):
            for l in range(3):
                assert zlib.decompress(streams_f[b][l]) == enc.layer(b, l)["coeffs"].tobytes(), f"{space} image {b} layer {l}: foreign exact table"
        blobs_gpu = codec.compress_many(batch, extension=".png", entropy="gpu")
        blobs_ref = codec.compress_many(batch, extension=".png")
        for b in range(batch.shape[0]):
            got = A.Jpeg(A.JpegCompressionSettings()).decompress(blobs_gpu[b]).data
            want = A.Jpeg(A.JpegCompressionSettings()).decompress(blobs_ref[b]).data
            assert np.array_equal(got, want)
            assert np.array_equal(oracle.decode_image(blobs_gpu[b]), oracle.decode_image(blobs_ref[b]), equal_nan=True)
    # natural image: the matcher works at arbitrary distances (token histogram of the parse) and the streams are close to zlib level 9's
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    enc = codec.compress_batch(golden_image("natural/baboon")[None])
    ctx = codec._bind()
    t = ctx.torch
    p = enc.plan
    nbytes = int(ctx.lib.aej_deflate_workspace_bytes(ctx.handle, 1, p.H, p.W))
    ws = ctx.workspace(nbytes)
    hist = ctx.empty((3, DT.HIST_BINS), t.int32)
    import ctypes
    ctx.check(ctx.lib.aej_deflate_histogram(ctx.handle, enc.coeffs.data_ptr(), enc.counts.data_ptr(), 1, p.H, p.W, hist.data_ptr(), ws.data_ptr(),
                                            ctypes.c_uint64(nbytes)))
    h = hist.cpu().numpy()
    dist = h[0, 286:316]
    assert dist[:4].sum() > 0 and dist[4:10].sum() > 0 and dist[10:18].sum() > 0 and dist[18:].sum() > 0, dist      # <= 4, <= 32, <= 512, beyond
    assert dist[14:].sum() > 0.2 * dist.sum(), dist                      # a fifth of the matches come from more than 128 bytes back
    assert h[0, 256] == 1 and h[0, :256].sum() > 0
    gpu_bytes = sum(len(x) for x in codec.deflate_batch(enc)[0])
    z9 = sum(len(zlib.compress(enc.layer(0, l)["coeffs"].tobytes(), 9)) for l in range(3))
    z1 = sum(len(zlib.compress(enc.layer(0, l)["coeffs"].tobytes(), 1)) for l in range(3))
    assert gpu_bytes < 1.2 * z9 and gpu_bytes < z1, (gpu_bytes, z9, z1)
    # Jpeg.compress's keyword-only opt-ins: same container layout, decoded by the default reader
    img = A.Image.from_array(golden_image("natural/house"), None, ".png")
    codec = A.Jpeg(A.JpegCompressionSettings("YCoCg", (40, 80), (4, 64)))
    ref_bytes = codec.compress(img)
    for kw in ({"entropy": "gpu"}, {"entropy": "gpu-fixed"}, {"zlib_level": 1}):
        alt = codec.compress(img, **kw)
        assert alt != ref_bytes and np.array_equal(A.Jpeg(A.JpegCompressionSettings()).decompress(alt).data,
                                                   A.Jpeg(A.JpegCompressionSettings()).decompress(ref_bytes).data), kw
    with pytest.raises(ValueError):
        codec.compress(img, entropy="cpu")
    # a stream slot that is too small is refused, not overrun; a count beyond its layer's capacity is refused, not followed
    from adaptive_edge_aware_jpeg_amd._lib import get_context
    ctx = get_context()
    enc = codec.compress_batch(img.data[None])
    p = enc.plan
    t = ctx.torch
    streams, sizes = ctx.empty((3, 256), t.uint8), ctx.empty((3,), t.int64)
    nbytes = int(ctx.lib.aej_deflate_workspace_bytes(ctx.handle, 1, p.H, p.W))
    ws = ctx.workspace(nbytes)
    rc = ctx.lib.aej_deflate_batch(ctx.handle, enc.coeffs.data_ptr(), enc.counts.data_ptr(), 1, p.H, p.W, None, 0, streams.data_ptr(),
                                   ctypes.c_uint64(256), sizes.data_ptr(), ws.data_ptr(), ctypes.c_uint64(nbytes))
    assert rc == -4 and int(sizes.cpu()[0]) > 256          # AEJ_ERR_CAPACITY, and the size it would have needed is reported
    bad_counts = enc.counts.clone()
    bad_counts[0, 1, 0] = 1 << 40
    cap = max((p.coeff_off[l + 1] if l < 2 else p.coeff_stride) - p.coeff_off[l] for l in range(3))
    stride = (int(ctx.lib.aej_deflate_stream_bound(ctypes.c_uint64(4 * cap))) + 255) // 256 * 256
    streams = ctx.empty((3, stride), t.uint8)
    rc = ctx.lib.aej_deflate_batch(ctx.handle, enc.coeffs.data_ptr(), bad_counts.data_ptr(), 1, p.H, p.W, None, 0, streams.data_ptr(),
                                   ctypes.c_uint64(stride), sizes.data_ptr(), ws.data_ptr(), ctypes.c_uint64(nbytes))
    assert rc == -1                                          # AEJ_ERR_ARG
    # an empty layer cannot occur in the codec, but the stream format must still close: one image of the smallest shape the settings allow
    tiny = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (2, 2)))
    enc = tiny.compress_batch(np.full((1, 4, 4, 3), 0.25, np.float32))
    for l in range(3):
        assert zlib.decompress(tiny.deflate_batch(enc)[0][l]) == enc.layer(0, l)["coeffs"].tobytes()


@pytest.mark.gpu
def test_sides_above_65535_pixels_are_refused_not_wrapped():
    """Leaf origins travel to the DCT kernels as 16-bit pairs (LeafWork, aej_common.h): a plan for a larger image is refused with
    AEJ_ERR_UNSUPPORTED (NotImplementedError here) instead of being computed with wrapped coordinates; 65535 itself is planned."""
    import adaptive_edge_aware_jpeg_amd as A
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    ctx = codec._bind()
    for H, W in ((65536, 16), (16, 70000)):
        with pytest.raises(NotImplementedError):
            ctx.plan(1, H, W)
    p = ctx.plan(1, 16, 65535 // 4 * 4)
    assert p.W == 65535 // 4 * 4 and p.workspace_bytes > 0


def test_compress_batches_keeps_calls_in_flight_and_yields_in_order(A, oracle):
    """Jpeg.compress_batches: a sequence of batches through up to `in_flight` calls on private streams (aej_encode_batch_begin / _end) -- every
    result equals compress_batch of the same input, results come in input order, inputs of different shapes and kinds (device tensor, numpy,
    uint8) mix, a consumer that stops early leaves nothing in flight, and a second pass reuses the streams and contexts of the first."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    imgs = [oracle.synth_image(160, 224, s, k).astype(np.float32) / np.float32(255.0) for s, k in ((1, "mixed"), (2, "noise"), (3, "flat"), (4, "mixed"))]
    batches = [np.stack(imgs[:2]), torch.from_numpy(np.stack(imgs[1:4])).to(dev), (np.stack(imgs[2:]) * 255).astype(np.uint8),
               torch.from_numpy(rng.random((1, 96, 128, 3), dtype=np.float32)).to(dev), np.stack(imgs[:1]), torch.from_numpy(np.stack(imgs)).to(dev)]
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    want = [codec.compress_batch(b) for b in batches]
    for in_flight in (3, 1, 4):
        got = list(codec.compress_batches(iter(batches), in_flight=in_flight))
        assert len(got) == len(want)
        for i, (g, w) in enumerate(zip(got, want)):
            assert torch.equal(g.counts, w.counts), (in_flight, i)
            for b in range(batches[i].shape[0]):
                for l in range(3):
                    a, c = g.layer(b, l), w.layer(b, l)
                    assert a["root_size"] == c["root_size"] and all(np.array_equal(a[k], c[k]) for k in ("states", "leaves", "coeffs")), (in_flight, i, b, l)
    n_streams = len(codec._pipe_streams)
    gen = codec.compress_batches(iter(batches), in_flight=3)
    first = next(gen)
    gen.close()                                            # the consumer stops: the calls in flight are ended, not abandoned
    assert torch.equal(first.counts, want[0].counts)
    assert len(codec._pipe_streams) == n_streams == 4
    again = list(codec.compress_batches(batches[:2], in_flight=2))
    for b in range(batches[1].shape[0]):                   # (whole buffers are not comparable: capacity beyond a layer's count is uninitialised)
        for l in range(3):
            assert np.array_equal(again[1].layer(b, l)["coeffs"], want[1].layer(b, l)["coeffs"])
    with pytest.raises(ValueError):
        list(codec.compress_batches(batches, in_flight=0))
    with pytest.raises(ValueError):
        list(codec.compress_batches([np.zeros((2, 8, 8), np.float32)]))
