"""CPU: the oracle against the real third-party libraries that ARE present (NumPy), plus algebraic
properties of the restated OpenCV stages (cv2 itself is absent: those stay 'parity unpinned')."""
import os

import numpy as np
import pytest


def test_matrix_colour_equals_np_dot(oracle):
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, size=(100000, 3)).astype(np.float32) / np.float32(255.0)
    M = np.array([[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5], [0.5, -0.418688, -0.081312]], dtype=np.float32)
    assert np.array_equal(oracle.color_forward("YCbCr", x), np.dot(x, M.T))     # ycbcr.py:61


def test_pow_accuracy(oracle):
    rng = np.random.default_rng(0)
    xs = rng.uniform(1e-7, 1.2, 200000)
    for y in (2.4, float(np.float32(1 / 3)), 2610.0 / 16384.0):
        rel = np.abs(oracle.pow_array(xs, y) - np.power(xs, y)) / np.power(xs, y)
        assert rel.max() < 1e-14
    assert oracle.pow_array(np.array([0.0]), 2.4)[0] == 0.0
    assert np.isnan(oracle.pow_array(np.array([-1.0]), 2.4)[0])


def test_oklab_cube_root_sequence_against_the_general_pow(oracle):
    """the oracle's OKLAB cube root (pow_third_f32, aej_oracle.c) is the operation sequence the HIP kernel shares, so it is
    pinned here against paths that share nothing with it: the file's general float64 pow rounded to float32, and NumPy's float64 power --
    over every lms value 8-bit images can reach (sRGB -> linear -> XYZ -> LMS of all grey levels and of 200 000 random colours), a dense
    sweep of the float32 range the seed is valid for, and the guards (0, tiny, huge, negative).  At most one float32 ulp anywhere, and the
    results must agree with the correctly rounded value in all but a small share of the inputs."""
    t = np.float64(np.float32(1.0 / 3.0))
    rng = np.random.default_rng(12)
    rgb = np.concatenate([np.repeat(np.arange(256, dtype=np.float32)[:, None], 3, 1), rng.integers(0, 256, (200000, 3)).astype(np.float32)]) / np.float32(255)
    xyz = oracle.color_forward("XYZ", rgb)
    M = np.array([[0.8189330101, 0.3618667424, -0.1288597137], [0.0329845436, 0.9293118715, 0.0361456387], [0.0482003018, 0.2643662691, 0.6338517070]], np.float32)
    lms = (xyz.astype(np.float32) @ M.T).astype(np.float32).ravel()
    lms = lms[lms > 0]
    sweep = np.exp(rng.uniform(np.log(1e-30), np.log(1e30), 400000)).astype(np.float32)
    for x in (lms, sweep):
        fast = oracle.pow_third_array(x)
        indep = oracle.pow_third_array(x, independent=True)
        exact = np.power(x.astype(np.float64), t)
        correctly_rounded = exact.astype(np.float32)
        ulp = np.spacing(np.abs(correctly_rounded))
        assert np.abs(fast.astype(np.float64) - exact).max() / np.abs(exact).max() < 1 and (np.abs(fast - correctly_rounded) <= ulp).all()
        assert (np.abs(indep - correctly_rounded) <= ulp).all()
        assert (fast != correctly_rounded).mean() < 0.01 and (indep != correctly_rounded).mean() < 1e-4
        assert (np.abs(fast - indep) <= ulp).all()
    guards = np.array([0.0, 1e-38, 3e-39, 1e-31, 1e31, 3e38, 1.0, 8.0, 27.0], np.float32)
    f, i = oracle.pow_third_array(guards), oracle.pow_third_array(guards, independent=True)
    assert f[0] == 0.0 and i[0] == 0.0 and (np.abs(f[1:] - i[1:]) <= np.spacing(np.abs(i[1:]))).all()
    assert np.isnan(oracle.pow_third_array(np.array([-1.0], np.float32))[0])
    # the whole OKLAB transform through the independent pow: within one ulp of the cube roots' effect on the outputs
    oracle.set_oklab_independent_pow(True)
    try:
        ref = oracle.color_forward("OKLAB", rgb)
    finally:
        oracle.set_oklab_independent_pow(False)
    assert np.abs(oracle.color_forward("OKLAB", rgb) - ref).max() <= 3e-7


def test_uint8_scaling_wraps_like_numpy_on_x86(oracle):
    v = np.array([-102.0, -2.55, 306.0, 0.0, 254.999, 255.0, 1.0], dtype=np.float32) / np.float32(255.0)
    got = oracle.to_u8(v)
    assert got[:3].tolist() == [154, 254, 50]          # SURVEY.md 8a-3 (measured with NumPy on x86)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(got, (v * 255).astype(np.uint8))


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 518400])
def test_percentiles_equal_numpy(oracle, n):
    rng = np.random.default_rng(n)
    for trial in range(4):
        img = np.clip(rng.normal(100 + 20 * trial, 5 + 20 * trial, size=n), 0, 255).astype(np.uint8)
        lo, hi = oracle.percentiles(img)
        assert lo == np.percentile(img, 0.10 * 100) and hi == np.percentile(img, 0.30 * 100)   # edge_detection.py:81-82


def test_canny_threshold_rule(oracle):
    assert oracle.canny_thresholds(37.0, 53.5) == (1369, 2862)
    assert oracle.canny_thresholds(53.5, 37.0) == (1369, 2862)        # swapped
    assert oracle.canny_thresholds(0.0, 40000.0) == (0, 32767 * 32767)


def test_reflect_pad_and_quantise_equal_numpy(oracle):
    """blocks_encode with unit quantisers and identity zigzag == np.pad(reflect) followed by our DCT; with the
    DCT replaced by identity-free checks: compare the gathered block through a 1x1 ... use s=4 leaves on a
    5x3 plane so both axes need multi-period reflection (jpeg.py:399-402)."""
    rng = np.random.default_rng(5)
    plane = rng.normal(0, 50, size=(3, 5)).astype(np.float32)
    s = 8
    leaves = np.array([[0, 0, s], [4, 0, s], [0, 2, s]], np.int32)
    ones = {s: np.ones((s, s), np.int32)}
    ident = {s: np.arange(s * s, dtype=np.int32)}
    _, dct = oracle.blocks_encode(plane, leaves, ones, ident, want_dct=True)
    D = oracle.dct_matrix(s).astype(np.float64)
    for i, (x, y, _) in enumerate(leaves):
        blk = plane[y:y + s, x:x + s]
        blk = np.pad(blk, ((0, s - blk.shape[0]), (0, s - blk.shape[1])), mode="reflect")
        ref = D @ blk.astype(np.float64) @ D.T
        assert np.abs(dct[i * s * s:(i + 1) * s * s].reshape(s, s) - ref).max() < 1e-3
    # np.round(f32 / int32) in float64, half-to-even (jpeg.py:501)
    y = np.array([2.5, 3.5, -2.5, 7.0, 1e-3, -0.4999], np.float32)
    q = np.array([1, 1, 1, 2, 1, 1], np.int32)
    assert np.round(y / q).astype(np.int32).tolist() == [2, 4, -2, 4, 0, 0]


@pytest.mark.parametrize("s", [4, 8, 16, 32, 64, 128])
def test_dct_contract_tolerance(oracle, s):
    """pre-quantisation DCT: float32 fma chains vs float64 orthonormal DCT-II; stated tolerance 1e-6 * s * 127."""
    rng = np.random.default_rng(s)
    blk = (rng.random((s, s), dtype=np.float32) * 254 - 127).astype(np.float32)
    ones = {s: np.ones((s, s), np.int32)}
    ident = {s: np.arange(s * s, dtype=np.int32)}
    co, d = oracle.blocks_encode(blk, np.array([[0, 0, s]], np.int32), ones, ident, want_dct=True)
    k = np.arange(s)
    C = np.sqrt(2.0 / s) * np.cos(np.pi * (2 * k[None, :] + 1) * k[:, None] / (2.0 * s))
    C[0, :] = np.sqrt(1.0 / s)
    ref = C @ blk.astype(np.float64) @ C.T
    assert np.abs(d.reshape(s, s) - ref).max() <= 1e-6 * s * 127
    assert np.abs(oracle.dct_matrix(s).astype(np.float64) @ oracle.dct_matrix(s).astype(np.float64).T - np.eye(s)).max() < 1e-6
    assert np.array_equal(co, np.rint(d.astype(np.float64)).astype(np.int32))


def test_stage_invariants_on_constant_and_simple_images(oracle):
    flat = np.full((40, 52), 77, np.uint8)
    assert np.all(oracle.gauss3(flat) == 77) and np.all(oracle.bilateral5(flat) == 77)
    assert oracle.canny(flat, 10.0, 30.0).sum() == 0
    # vertical step edge: one column of edge pixels, full height
    step = np.zeros((32, 32), np.uint8)
    step[:, 16:] = 200
    e = oracle.canny(step, 50.0, 100.0)
    cols = np.nonzero(e.any(axis=0))[0]
    assert len(cols) == 1 and e[:, cols[0]].all()
    # CLAHE keeps a constant image constant up to its LUT value and never leaves [0,255]
    c = oracle.clahe(flat)
    assert c.min() == c.max()
    # Gaussian reflect-101 border: a single bright pixel in the corner spreads 4/16 + 2*(2/16)*... = (4+2+2+1)/16 weight
    img = np.zeros((8, 8), np.uint8)
    img[0, 0] = 160
    assert oracle.gauss3(img)[0, 0] == (4 * 160 + 8) >> 4


def test_edge_pipeline_shapes_and_determinism(oracle):
    img = oracle.synth_image(67, 101, 3).astype(np.float32) / np.float32(255.0)
    plane = oracle.color_forward("YCbCr", img.reshape(-1, 3)).reshape(67, 101, 3)[:, :, 0].copy()
    e1, st, thr = oracle.edge_pipeline(plane, return_stages=True)
    e2 = oracle.edge_pipeline(plane)
    assert np.array_equal(e1, e2) and set(np.unique(e1)) <= {0, 1}
    assert st.shape == (4, 67, 101)
    assert (thr[0], thr[1]) == oracle.percentiles(st[3])


def test_quadtree_structure_properties(oracle):
    """leaves tile the image exactly once; the reference's own header decoder recovers the leaf sizes."""
    rng = np.random.default_rng(11)
    for (h, w, mn, mx) in ((150, 211, 4, 64), (64, 64, 8, 8), (300, 77, 4, 128), (9, 9, 2, 4)):
        edge = (rng.random((h, w)) < 0.01).astype(np.uint8)
        leaves, states, root = oracle.quadtree(edge, mn, mx)
        cover = np.zeros((h, w), np.int32)
        for x, y, s in leaves:
            cover[y:y + s, x:x + s] += 1
        assert np.all(cover == 1)
        sizes, stack, i = [], [root], 0
        while stack and i < len(states):
            size = stack.pop()
            st = states[i]
            i += 1
            if st == 0:
                sizes.append(size)
            elif st == 1:
                stack.extend([size // 2] * 4)
        assert sizes == leaves[:, 2].tolist()


def test_general_inter_area_against_float64_integration(oracle):
    """odd sizes: cv.resize(INTER_AREA) is an area-weighted mean over [d*scale, (d+1)*scale) (clipped at the image edge)."""
    rng = np.random.default_rng(0)

    def area_ref(a, Ho, Wo):
        H, W = a.shape
        sy, sx = H / Ho, W / Wo
        out = np.zeros((Ho, Wo))
        for y in range(Ho):
            for x in range(Wo):
                y0, y1, x0, x1 = y * sy, min((y + 1) * sy, H), x * sx, min((x + 1) * sx, W)
                acc = area = 0.0
                for yy in range(int(np.floor(y0)), int(np.ceil(y1))):
                    wy = min(yy + 1, y1) - max(yy, y0)
                    for xx in range(int(np.floor(x0)), int(np.ceil(x1))):
                        wx = min(xx + 1, x1) - max(xx, x0)
                        acc += a[yy, xx] * wy * wx
                        area += wy * wx
                out[y, x] = acc / area
        return out

    for (H, W, rh, rw) in ((7, 9, 2, 2), (33, 35, 2, 2), (12, 50, 1, 4), (5, 9, 2, 2), (10, 10, 2, 2), (9, 8, 1, 4)):
        conv = rng.random((H, W, 3), dtype=np.float32)
        for ch in (0, 2):
            got = oracle.downsample(conv, ch, rh, rw)
            assert got.shape == (H // rh, W // rw)
            assert np.abs(got - area_ref(conv[:, :, ch].astype(np.float64), H // rh, W // rw)).max() < 5e-7
    flat = np.full((11, 13, 3), 0.3, np.float32)
    assert np.abs(oracle.downsample(flat, 1, 2, 2) - np.float32(0.3)).max() < 1e-7
    same = oracle.downsample(rng.random((6, 6, 3), dtype=np.float32), 0, 1, 1)
    assert same.shape == (6, 6)


def test_reference_structured_restatement_equals_the_oracle(oracle):
    """oracle/reference_structured.py keeps the reference's control structure (explicit-stack quadtree with one region test per
    node, per-leaf Python loops for pad / DCT / quantise / zigzag: quadtree.py:93-165, jpeg.py:393-404,471,499-502,581-588);
    it is bench.py's second CPU baseline.  Same bytes as the one-C-call-per-stage oracle, ragged shapes included."""
    from oracle import reference_structured as R
    for (H, W, sp, br) in ((150, 211, "YCbCr", (4, 64)), (67, 101, "OKLAB", (4, 32)), (64, 100, "ICtCp", (8, 8)), (5, 9, "YCoCg", (2, 4))):
        img = oracle.synth_image(H, W, H + W).astype(np.float32) / np.float32(255.0)
        a, b = oracle.encode_image(img, sp, (40, 80), br), R.encode_image(img, sp, (40, 80), br)
        for l in range(3):
            assert a[l]["root_size"] == b[l]["root_size"]
            for k in ("states", "leaves", "coeffs"):
                assert np.array_equal(a[l][k], b[l][k]), (sp, l, k)


def test_oracle_runs_clean_under_address_and_ub_sanitizers(oracle):
    """The C oracle is the checker behind every parity claim: run its whole encode + decode path (ragged shapes, every colour
    space, block sizes 2..128, overhanging leaves) in a child interpreter against the -fsanitize=address,undefined build, with
    libasan preloaded.  Any out-of-bounds access, use-after-free or undefined shift / overflow aborts the child."""
    import subprocess
    import sys
    from conftest import ROOT
    oracle.build(sanitize=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found next to gcc")
    code = (
        "import sys; sys.path.insert(0, sys.argv[1])\n"
        "import numpy as np\n"
        "from oracle import oracle as O\n"
        "cases = [(67, 101, 'YCbCr', (4, 64)), (33, 50, 'OKLAB', (4, 128)), (40, 51, 'ICtCp', (2, 16)), (5, 9, 'YCoCg', (2, 4)),\n"
        "         (130, 260, 'JzAzBz', (8, 32)), (64, 62, 'ICaCb', (4, 16)), (96, 160, 'YCoCg-R', (16, 16))]\n"
        "for (H, W, sp, br) in cases:\n"
        "    img = O.synth_image(H, W, H * W).astype(np.float32) / np.float32(255.0)\n"
        "    layers = O.encode_image(img, sp, (40, 80), br)\n"
        "    data = O.write_ajpg(layers, H, W, sp, (40, 80), br, '.png')\n"
        "    out = O.decode_image(data)\n"
        "    assert out.shape == (H, W, 3)\n"
        "e = (np.random.default_rng(1).random((150, 90)) < 0.01).astype(np.uint8)\n"
        "O.quadtree(e, 2, 256); O.quadtree(e[:1, :1], 4, 64)\n"
        "print('clean')\n")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               AEJ_ORACLE_SANITIZE="1")
    r = subprocess.run([sys.executable, "-c", code, ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("clean"), (r.stdout[-500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_oracle_stencils_against_independent_scipy_restatements(oracle):
    """OpenCV itself is absent (parity unpinned), but the oracle's stencil stages can at least be checked against
    restatements that share no code with it: SciPy's convolution / resampling with the border rules OpenCV documents.
    * GaussianBlur 3x3, sigma 0 on uint8 = ([1 2 1]^T [1 2 1] * p + 8) >> 4 with BORDER_REFLECT_101 (scipy mode 'mirror');
    * Sobel 3x3 with BORDER_REPLICATE (scipy mode 'nearest'): every STRONG pixel of the NMS map has mag = dx^2 + dy^2 > high, every
      pixel with mag <= low is suppressed (the NMS in between is the oracle's own);
    * cv.resize INTER_AREA with integer ratio 2 x 2 = the mean of each 2 x 2 block."""
    from scipy import ndimage
    rng = np.random.default_rng(11)
    img = (rng.random((97, 131)) * 255).astype(np.uint8)
    img[20:60, 30:90] = 200                                  # a flat patch and its edges
    k = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], np.int64)
    want = ((ndimage.convolve(img.astype(np.int64), k, mode="mirror") + 8) >> 4).astype(np.uint8)
    assert np.array_equal(oracle.gauss3(img), want)
    lo, hi = 40.0, 90.0
    edge, nms = oracle.canny(img, lo, hi, return_nms=True)
    p = img.astype(np.int64)
    dx = ndimage.correlate(p, np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.int64), mode="nearest")
    dy = ndimage.correlate(p, np.array([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], np.int64), mode="nearest")
    mag = dx * dx + dy * dy
    low, high = oracle.canny_thresholds(lo, hi)
    assert (mag[nms == 2] > high).all() and (nms[mag <= low] == 1).all()
    assert set(np.unique(edge).tolist()) <= {0, 255} and (edge[nms == 2] == 255).all() and (edge[nms == 1] == 0).all()      # cv.Canny's 0 / 255 map
    conv = rng.random((3, 40, 52)).astype(np.float32)
    hw3 = np.ascontiguousarray(conv.transpose(1, 2, 0))
    got = oracle.downsample(hw3, 1, 2, 2)
    blocks = conv[1].reshape(20, 2, 26, 2)
    want = ((blocks[:, 0, :, 0] + blocks[:, 0, :, 1]) + (blocks[:, 1, :, 0] + blocks[:, 1, :, 1])) * np.float32(0.25)
    assert np.array_equal(got, want.astype(np.float32))
