"""GPU (-m gpu): BASELINE.json's full-size configurations against the CPU oracle, bit for bit.

* config 4's image class: one 3840x2160 image, YCbCr, blocks 4-64;
* config 5's image class: one 7680x4320 image, OKLAB, blocks 4-128 (root 8192: 13-bit Morton codes, pyramid levels >= 5);
* the batched layouts the bench times: B = 8 x 4K and B = 64 x 1080p (int64 plane offsets, per-plane work lists, kMaxPlanes
  prefix tables), images 0, B/2 and B-1 compared with the oracle;
* determinism: the same batch encoded twice in one process gives identical bytes (the hysteresis chase passes race by design;
  the fix-point must not depend on the interleaving);
* contexts follow torch's current stream.

Images come from bench.synth_batch (the generator the bench times) so that the tested inputs are the benchmarked ones; the
oracle runs on their host copies (C, about 1.5 s per 4K image and 11 s for the 8K one).
"""
import hashlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import adaptive_edge_aware_jpeg_amd as pkg
    import bench
    return torch, pkg, bench


def check_image(enc, b, ref, tag):
    for l in range(3):
        got = enc.layer(b, l)
        assert got["root_size"] == ref[l]["root_size"], f"{tag} L{l} root"
        assert np.array_equal(got["states"], ref[l]["states"]), f"{tag} L{l} states"
        assert np.array_equal(got["leaves"], ref[l]["leaves"]), f"{tag} L{l} leaves"
        assert np.array_equal(got["coeffs"], ref[l]["coeffs"]), f"{tag} L{l} coeffs"


@pytest.mark.parametrize("H,W,space,br", [(2160, 3840, "YCbCr", (4, 64)), (4320, 7680, "OKLAB", (4, 128)), (2160, 3840, "ICtCp", (4, 64)),
                                          (2160, 3840, "JzAzBz", (8, 128)), (2160, 3840, "ICaCb", (4, 32)), (2160, 3840, "YCoCg-R", (2, 64))],
                         ids=["4K-YCbCr-4-64", "8K-OKLAB-4-128", "4K-ICtCp-4-64", "4K-JzAzBz-8-128", "4K-ICaCb-4-32", "4K-YCoCg-R-2-64"])
def test_single_full_size_image_matches_oracle(env, oracle, H, W, space, br):
    torch, A, bench = env
    x = bench.synth_batch(torch, 1, H, W, 20250718, torch.device("cuda", 0))
    enc = A.Jpeg(A.JpegCompressionSettings(space, (40, 80), br)).compress_batch(x)
    ref = oracle.encode_image(x[0].cpu().numpy(), space, (40, 80), br)
    check_image(enc, 0, ref, f"{W}x{H} {space}")
    sizes = set(int(s) for l in range(3) for s in np.unique(enc.layer(0, l)["leaves"][:, 2]))
    assert min(sizes) == br[0] and max(sizes) == br[1], sizes          # the whole block range occurs in the image


@pytest.mark.parametrize("B,H,W,space,br", [(8, 2160, 3840, "YCbCr", (4, 64)), (64, 1080, 1920, "YCbCr", (4, 64)), (8, 4320, 7680, "OKLAB", (4, 128))],
                         ids=["8x4K", "64x1080p", "8x8K-OKLAB-4-128"])
def test_batched_layout_matches_oracle(env, oracle, B, H, W, space, br):
    """The batched layouts of BASELINE configs 3-5 (config 5's per-GPU share is the 8 x 7680x4320 OKLAB 4-128 case: images 0 and 7 against
    the oracle, about 11 s each on two threads)."""
    torch, A, bench = env
    qr = (40, 80)
    x = bench.synth_batch(torch, B, H, W, 20250718, torch.device("cuda", 0))
    codec = A.Jpeg(A.JpegCompressionSettings(space, qr, br))
    enc = codec.compress_batch(x)
    picks = sorted({0, B // 2, B - 1}) if H < 4320 else [0, B - 1]
    with ThreadPoolExecutor(max_workers=len(picks)) as ex:         # ctypes releases the GIL inside the C oracle
        refs = list(ex.map(lambda b: oracle.encode_image(x[b].cpu().numpy(), space, qr, br), picks))
    for b, ref in zip(picks, refs):
        check_image(enc, b, ref, f"image {b} of {B}")
    # every image of the batch: counts consistent with the per-layer tables
    cnt = enc.counts_host
    for b in range(B):
        for l in range(3):
            n_coef, n_leaf, n_state, root = (int(v) for v in cnt[b, l])
            assert root == enc.plan.root_size[l] and n_leaf > 0 and n_state >= n_leaf and n_coef >= enc.plan.layer_h[l] * enc.plan.layer_w[l]


def digest(enc, B):
    h = hashlib.sha256()
    h.update(enc.counts_host.tobytes())
    for b in range(B):
        for l in range(3):
            d = enc.layer(b, l)
            for k in ("coeffs", "leaves", "states"):
                h.update(d[k].tobytes())
    return h.hexdigest()


def test_encode_is_deterministic_in_process(env):
    """the waves that drain the hysteresis work queue write with atomics while other waves read; the result must not depend
    on the interleaving.  Same device-resident batch, three encodes, identical digests."""
    torch, A, bench = env
    B, H, W = 8, 2160, 3840
    x = bench.synth_batch(torch, B, H, W, 4242, torch.device("cuda", 0))
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    d = [digest(codec.compress_batch(x), B) for _ in range(3)]
    assert d[0] == d[1] == d[2]


def test_hysteresis_queue_statistics(env, oracle):
    """The hysteresis completes on the device (a pass over every tile, then the work queue drained by one persistent launch): repeated
    calls give the same bytes, and the context reports how many tiles went through the queue -- some, but a small fraction."""
    torch, A, bench = env
    from adaptive_edge_aware_jpeg_amd._lib import get_context
    B, H, W = 6, 1080, 1920
    dev = torch.device("cuda", 0)
    x1 = bench.synth_batch(torch, B, H, W, 1, dev)
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    ctx = get_context()
    s0 = ctx.hysteresis_stats()
    a = digest(codec.compress_batch(x1), B)
    b = digest(codec.compress_batch(x1), B)
    s1 = ctx.hysteresis_stats()
    assert a == b and s1["calls"] - s0["calls"] == 2
    tiles = B * (17 * 30 + 2 * 9 * 15)
    assert 0 < s1["queued"] < tiles, (s1, tiles)
    check_image(codec.compress_batch(x1), B - 1, oracle.encode_image(x1[B - 1].cpu().numpy(), "YCbCr", (40, 80), (4, 64)), "last image")


def test_contexts_follow_the_current_stream(env, oracle):
    """work issued under `torch.cuda.stream(s)` must run on s -- inputs produced on s are read by our
    kernels without any cross-stream hazard, and each stream gets its own context / workspace."""
    torch, A, bench = env
    from adaptive_edge_aware_jpeg_amd._lib import get_context
    dev = torch.device("cuda", 0)
    img = oracle.synth_image(256, 384, 77).astype(np.float32) / np.float32(255.0)
    ref = oracle.encode_image(img, "YCbCr", (40, 80), (4, 64))
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    c0 = get_context()
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        c1 = get_context()
        assert c1 is not c0 and c1.stream == s.cuda_stream
        big = torch.zeros((64, 1024, 1024), device=dev)         # keeps s busy so that the producer really is "in flight"
        for _ in range(4):
            big += 1.0
        x = torch.from_numpy(img[None]).to(dev, non_blocking=True) + (big[0, :1, :1, None] * 0.0)   # produced ON s, after the busy work
        enc = codec.compress_batch(x)
        with pytest.raises(Exception):
            c0.workspace(16)                                     # a context bound to another stream refuses to serve
    s.synchronize()
    check_image(enc, 0, ref, "side stream")
    assert get_context() is c0
    # aej_set_stream (C ABI) re-binds a context explicitly
    c1.check(c1.lib.aej_set_stream(c1.handle, None))
    c1.check(c1.lib.aej_set_stream(c1.handle, s.cuda_stream))


def test_xyz_helper_space_and_float64_input(env, oracle):
    import os
    from conftest import GOLDEN
    torch, A, bench = env
    g = np.load(os.path.join(GOLDEN, "color_xyz.npz"))
    x = g["rgb_u8"].astype(np.float32) / np.float32(255.0)
    assert np.array_equal(A.convert("sRGB", "XYZ", x), g["XYZ"])                 # the reference's own output
    assert np.array_equal(A.convert("XYZ", "sRGB", g["XYZ_in"]), g["sRGB"])
    assert np.array_equal(A.apply_normalization("XYZ", g["XYZ"], False).astype(np.float32), g["XYZ_norm"])
    y64 = A.convert("sRGB", "YCbCr", x.astype(np.float64))                       # float64 in: rounded to float32, float32 out
    assert y64.dtype == np.float32 and np.array_equal(y64, A.convert("sRGB", "YCbCr", x))
    with pytest.raises(ValueError):
        A.convert("XYZ", "YCbCr", x)
    with pytest.raises(ValueError):
        A.JpegCompressionSettings("XYZ")                                          # not a codec space (jpeg.py:164-165)


def test_graph_replay_path_matches_oracle(env, oracle):
    """Launch-latency path (BASELINE config 2, one 1920x1080 image): from the third call with the same buffers on, the whole launch
    sequence is one captured hipGraph; its outputs are the oracle's, also when the input changes between replays, for batches and for
    shapes whose workspace planes are tiled."""
    torch, A, bench = env
    from adaptive_edge_aware_jpeg_amd._lib import get_context
    dev = torch.device("cuda", 0)
    space, qr, br = "YCbCr", (40, 80), (4, 64)
    H, W = 1080, 1920
    xs = [bench.synth_batch(torch, 1, H, W, seed, dev) for seed in (20250718, 99)]
    refs = [oracle.encode_image(x[0].cpu().numpy(), space, qr, br) for x in xs]
    codec = A.Jpeg(A.JpegCompressionSettings(space, qr, br))
    ctx = codec._bind()
    ctx.set_graph_mode(2)
    try:
        plan = ctx.plan(1, H, W)
        out = (ctx.empty((plan.coeff_stride,), torch.int32), ctx.empty((plan.leaf_stride, 4), torch.int32),
               ctx.empty((plan.state_stride,), torch.uint8), ctx.empty((1, 3, 4), torch.int64))
        x = xs[0].clone()
        g0 = ctx.graph_stats()
        from adaptive_edge_aware_jpeg_amd.jpeg import EncodedBatch
        for it in range(6):
            which = it & 1
            x.copy_(xs[which])                          # same input buffer, new contents: the graph must pick them up
            codec.encode_into(ctx, x, plan, *out)
            check_image(EncodedBatch(plan, *out), 0, refs[which], f"graph call {it}")
        g1 = ctx.graph_stats()
        assert g1["captures"] - g0["captures"] == 1 and g1["launches"] - g0["launches"] >= 3, (g0, g1)
        _graph_replay_equals_eager(torch, bench, oracle, codec, ctx, dev, 6, H, W, space, qr, br)
        # a shape whose workspace planes are kept in 4 x 4 blocks (3 x 4K float32 YCbCr: strips of 32 rows) -- capture
        # and replay must agree on the layout (the speculation-miss repair that could disagree with them no longer exists: the
        # hysteresis completes on the device)
        _graph_replay_equals_eager(torch, bench, oracle, codec, ctx, dev, 3, 2160, 3840, space, qr, br)
    finally:
        ctx.set_graph_mode(0)


def _graph_replay_equals_eager(torch, bench, oracle, codec, ctx, dev, B, H, W, space, qr, br):
    """a batch replayed through the graph gives the eager path's outputs, and image 0 the oracle's"""
    xb = bench.synth_batch(torch, B, H, W, 5, dev)
    planb = ctx.plan(B, H, W)

    def outs():
        return (ctx.empty((B * planb.coeff_stride,), torch.int32), ctx.empty((B * planb.leaf_stride, 4), torch.int32),
                ctx.empty((B * planb.state_stride,), torch.uint8), ctx.empty((B, 3, 4), torch.int64))
    ctx.set_graph_mode(0)
    want = outs()
    codec.encode_into(ctx, xb, planb, *want)
    torch.cuda.synchronize()
    ctx.set_graph_mode(2)
    got = outs()
    g2 = ctx.graph_stats()
    for it in range(3):                              # 1st: remembered, 2nd: captured + replayed, 3rd: replayed
        for t in got:
            t.zero_()
        codec.encode_into(ctx, xb, planb, *got)
        torch.cuda.synchronize()
        _assert_same_encoding(torch, planb, got, want, B, f"graph call {it}, {B} x {W}x{H}")
    assert ctx.graph_stats()["launches"] - g2["launches"] >= 2
    from adaptive_edge_aware_jpeg_amd.jpeg import EncodedBatch
    check_image(EncodedBatch(planb, *got), 0, oracle.encode_image(xb[0].cpu().numpy(), space, qr, br), f"graph replay {B} x {W}x{H}: image 0")


def _encode_raw(torch, ctx, codec, x, plan):
    out = (ctx.empty((x.shape[0] * plan.coeff_stride,), torch.int32), ctx.empty((x.shape[0] * plan.leaf_stride, 4), torch.int32),
           ctx.empty((x.shape[0] * plan.state_stride,), torch.uint8), ctx.empty((x.shape[0], 3, 4), torch.int64))
    for t in out:
        t.zero_()
    codec.encode_into(ctx, x, plan, *out)
    torch.cuda.synchronize()
    return out


def _assert_same_encoding(torch, plan, got, want, B, what):
    assert torch.equal(got[3], want[3]), f"{what}: counts"
    for im in range(B):
        for l in range(3):
            nc, nl, ns, _ = (int(v) for v in want[3][im, l])
            c0, l0, s0 = im * plan.coeff_stride + plan.coeff_off[l], im * plan.leaf_stride + plan.leaf_off[l], im * plan.state_stride + plan.state_off[l]
            assert torch.equal(got[0][c0:c0 + nc], want[0][c0:c0 + nc]), f"{what}: image {im} layer {l} coefficients"
            assert torch.equal(got[1][l0:l0 + nl], want[1][l0:l0 + nl]), f"{what}: image {im} layer {l} leaves"
            assert torch.equal(got[2][s0:s0 + ns], want[2][s0:s0 + ns]), f"{what}: image {im} layer {l} states"


def test_sub_batch_pipelining_is_invisible(env, oracle):
    """aej_set_sub_batches: a call cut into sub-batches on private streams (the throughput path of the 64 x 4K bench) returns what the
    unsplit call returns -- even and uneven splits, float and 8-bit ingest -- and image 0 / the last image equal the oracle."""
    torch, A, bench = env
    dev = torch.device("cuda", 0)
    space, qr, br = "YCbCr", (40, 80), (4, 64)
    B, H, W = 14, 1080, 1920
    x = bench.synth_batch(torch, B, H, W, 31, dev)
    codec = A.Jpeg(A.JpegCompressionSettings(space, qr, br))
    ctx = codec._bind()
    plan = ctx.plan(B, H, W)
    try:
        ctx.set_sub_batches(1)
        want = _encode_raw(torch, ctx, codec, x, plan)
        want2 = _encode_raw(torch, ctx, codec, x, plan)
        _assert_same_encoding(torch, plan, want2, want, B, "unsplit, second call")
        from adaptive_edge_aware_jpeg_amd.jpeg import EncodedBatch
        for im in (0, B - 1):
            check_image(EncodedBatch(plan, *want), im, oracle.encode_image(x[im].cpu().numpy(), space, qr, br), f"unsplit image {im}")
        for n in (2, 3, 4, 8):
            ctx.set_sub_batches(n)
            c0 = ctx.split_calls()
            got = _encode_raw(torch, ctx, codec, x, plan)
            assert ctx.split_calls() == c0 + 1
            _assert_same_encoding(torch, plan, got, want, B, f"{n} sub-batches")
        # 8-bit ingest through the split path (the GPU forms float32(v) / 255 exactly as the float batch was made)
        ctx.set_sub_batches(3)
        x8 = (x * 255.0).round().to(torch.uint8)
        got = _encode_raw(torch, ctx, codec, x8, plan)
        _assert_same_encoding(torch, plan, got, want, B, "3 sub-batches, uint8 ingest")
    finally:
        ctx.set_sub_batches(0)


def test_begin_end_on_two_contexts(env, oracle):
    """aej_encode_batch_begin / _end: two contexts on two streams with a call in flight each return what the blocking call returns;
    misuse (a second begin, settings while in flight, end without begin) is refused."""
    torch, A, bench = env
    from adaptive_edge_aware_jpeg_amd._lib import AejError
    dev = torch.device("cuda", 0)
    space, qr, br = "YCbCr", (40, 80), (4, 64)
    B, H, W = 3, 720, 1280
    xs = [bench.synth_batch(torch, B, H, W, seed, dev) for seed in (41, 42)]
    codec = A.Jpeg(A.JpegCompressionSettings(space, qr, br))
    ctx0 = codec._bind()
    plan = ctx0.plan(B, H, W)
    want = [_encode_raw(torch, ctx0, codec, x, plan) for x in xs]
    s1 = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s1):
        ctx1 = codec._bind()
        assert ctx1 is not ctx0
        out1 = (ctx1.empty((B * plan.coeff_stride,), torch.int32), ctx1.empty((B * plan.leaf_stride, 4), torch.int32),
                ctx1.empty((B * plan.state_stride,), torch.uint8), ctx1.empty((B, 3, 4), torch.int64))
    out0 = (ctx0.empty((B * plan.coeff_stride,), torch.int32), ctx0.empty((B * plan.leaf_stride, 4), torch.int32),
            ctx0.empty((B * plan.state_stride,), torch.uint8), ctx0.empty((B, 3, 4), torch.int64))
    torch.cuda.synchronize()
    for rnd in range(3):
        a, b = rnd & 1, (rnd + 1) & 1
        codec.encode_begin(ctx0, xs[a], plan, *out0)
        with torch.cuda.stream(s1):
            codec.encode_begin(ctx1, xs[b], plan, *out1)
        with pytest.raises(AejError):
            codec.encode_begin(ctx0, xs[a], plan, *out0)            # one call in flight per context
        with pytest.raises(AejError):
            ctx0.set_profiling(True)                                # nor any other entry point that uses the context's stream
        codec.encode_end(ctx0)
        with torch.cuda.stream(s1):
            codec.encode_end(ctx1)
        torch.cuda.synchronize()
        _assert_same_encoding(torch, plan, out0, want[a], B, f"round {rnd}, context 0")
        _assert_same_encoding(torch, plan, out1, want[b], B, f"round {rnd}, context 1")
    with pytest.raises(AejError):
        codec.encode_end(ctx0)                                      # nothing in flight
    ref = oracle.encode_image(xs[0][0].cpu().numpy(), space, qr, br)
    from adaptive_edge_aware_jpeg_amd.jpeg import EncodedBatch
    check_image(EncodedBatch(plan, *want[0]), 0, ref, "blocking call")


def test_randomised_parity_sweep(env, oracle):
    """A bounded, seeded slice of tests/manual/fuzz_gpu.py inside the suite: random sizes (ragged and aligned), colour spaces,
    block ranges, quality ranges and image kinds (8-bit levels and arbitrary floats) -- whole encode and decode against the oracle."""
    torch, A, bench = env
    rng = np.random.default_rng(20250718)
    spaces = ["YCbCr", "YCoCg", "YCoCg-R", "OKLAB", "ICtCp", "ICaCb", "JzAzBz"]
    for case in range(24):
        space = spaces[case % len(spaces)]
        H, W = int(rng.integers(5, 420)), int(rng.integers(5, 560))
        if rng.random() < 0.3:
            H, W = (H // 4 + 1) * 4, (W // 4 + 1) * 4
        lo = int(2 ** rng.integers(1, 5))
        hi = min(int(lo * 2 ** rng.integers(0, 5)), 256)
        q = tuple(sorted(int(v) for v in rng.integers(5, 96, size=2)))
        kind = ["mixed", "noise", "flat", "mixed"][rng.integers(4)]
        img = oracle.synth_image(H, W, int(rng.integers(1 << 30)), kind).astype(np.float32) / np.float32(255.0)
        if rng.random() < 0.3:            # arbitrary floats beside 8-bit levels
            img = np.where(rng.random((H, W, 1)) < 0.5, img, rng.random((H, W, 3), dtype=np.float32)).astype(np.float32)
        tag = f"case {case}: {space} {H}x{W} blocks ({lo},{hi}) q {q} {kind}"
        codec = A.Jpeg(A.JpegCompressionSettings(space, q, (lo, hi)))
        enc = codec.compress_batch(img[None])
        ref = oracle.encode_image(img, space, q, (lo, hi))
        check_image(enc, 0, ref, tag)
        dec = codec.decompress_batch(enc).cpu().numpy()[0]
        want = oracle.decode_image(oracle.write_ajpg(ref, H, W, space, q, (lo, hi), ".png"))
        assert np.array_equal(dec, want, equal_nan=True), tag + " decode"


def test_bench_configuration_64x4k_four_contexts(env, oracle):
    """The exact configuration bench.py times (BASELINE config 4's per-GPU share): B = 64 x 3840x2160, FOUR contexts on four streams
    with aej_encode_batch_begin / _end (bench.py --pipeline 4), every call cut into two sub-batches (bench.pipelined_sub_batches), two
    different input batches rotated with the bench's own input_of(); images 0, B/2 and B-1 of EACH context's last call are compared with
    the oracle, and every image's counters are checked for consistency."""
    torch, A, bench = env
    from adaptive_edge_aware_jpeg_amd.jpeg import EncodedBatch
    dev = torch.device("cuda", 0)
    space, qr, br = "YCbCr", (40, 80), (4, 64)
    B, H, W = 64, 2160, 3840
    n_pipe, n_steps = 4, 12
    codec = A.Jpeg(A.JpegCompressionSettings(space, qr, br))
    xs = [bench.synth_batch(torch, B, H, W, seed, dev) for seed in (20250718, 20250718 + 1_000_000)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_pipe)]
    ctxs, outs = [], []
    for s in streams:
        with torch.cuda.stream(s):
            c = codec._bind()
            c.set_sub_batches(2)
            plan = c.plan(B, H, W)
            ctxs.append(c)
            outs.append((c.empty((B * plan.coeff_stride,), torch.int32), c.empty((B * plan.leaf_stride, 4), torch.int32),
                         c.empty((B * plan.state_stride,), torch.uint8), c.empty((B, 3, 4), torch.int64)))
    torch.cuda.synchronize()
    # the test process asks for 16 hardware queues before HIP starts (tests/conftest.py), as bench.py does; state it in case the
    # process was started otherwise -- with fewer real queues the streams share queues, which is slower but equally correct
    for c in ctxs:
        c.check(c.lib.aej_set_hw_queues(c.handle, 16))
    assert ctxs[0].schedule(B, H, W)["sub_batches"] == 2

    def input_of(i):                              # bench.py's rotation
        return (i // n_pipe + i) & 1

    picks = [0, B // 2, B - 1]
    with ThreadPoolExecutor(max_workers=6) as ex:
        futs = {(w, b): ex.submit(oracle.encode_image, xs[w][b].cpu().numpy(), space, qr, br) for w in (0, 1) for b in picks}
        split0 = [c.split_calls() for c in ctxs]
        last = {}
        for i in range(n_steps):
            k, which = i % n_pipe, input_of(i)
            if i >= n_pipe:
                with torch.cuda.stream(streams[k]):
                    codec.encode_end(ctxs[k])
            with torch.cuda.stream(streams[k]):
                codec.encode_begin(ctxs[k], xs[which], plan, *outs[k])
            last[k] = which
        for k in range(n_pipe):
            with torch.cuda.stream(streams[k]):
                codec.encode_end(ctxs[k])
        torch.cuda.synchronize()
        assert 0 < sum(last.values()) < n_pipe, "both input batches must be among the contexts' last calls"
        assert all(c.split_calls() - s0 == n_steps // n_pipe for c, s0 in zip(ctxs, split0)), "the calls were expected to run as sub-batches"
        for k in range(n_pipe):
            enc = EncodedBatch(plan, *outs[k])
            for b in picks:
                check_image(enc, b, futs[(last[k], b)].result(), f"context {k}, batch {last[k]}, image {b} of {B}")
            cnt = enc.counts_host
            assert (cnt[:, :, 3] == np.asarray(plan.root_size)[None, :]).all() and (cnt[:, :, 1] > 0).all()
            assert (cnt[:, :, 0] >= (np.asarray(plan.layer_h) * np.asarray(plan.layer_w))[None, :]).all()
    from adaptive_edge_aware_jpeg_amd import hw_queues
    for c in ctxs:
        c.check(c.lib.aej_set_hw_queues(c.handle, hw_queues()[0]))


def test_failed_begin_drains_what_it_enqueued(env, oracle):
    """A whole-path call that fails after it has put work in flight (aej_test_fail_after_stage: test instrumentation of the ABI) must
    return with its streams drained -- unsplit and sub-batch paths alike -- leave the context usable, and leave no stale chain hook."""
    torch, A, bench = env
    from adaptive_edge_aware_jpeg_amd._lib import AejError
    from adaptive_edge_aware_jpeg_amd.jpeg import EncodedBatch
    dev = torch.device("cuda", 0)
    space, qr, br = "YCbCr", (40, 80), (4, 64)
    B, H, W = 14, 1080, 1920
    x = bench.synth_batch(torch, B, H, W, 31, dev)
    codec = A.Jpeg(A.JpegCompressionSettings(space, qr, br))
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        ctx = codec._bind()
        plan = ctx.plan(B, H, W)
        try:
            ctx.set_sub_batches(1)
            want = _encode_raw(torch, ctx, codec, x, plan)
            stages = {"color_planes": 1, "hysteresis": 6, "dct64": 13}
            for nsub in (1, 2):
                ctx.set_sub_batches(nsub)
                for name, stage in stages.items():
                    out = (ctx.empty((B * plan.coeff_stride,), torch.int32), ctx.empty((B * plan.leaf_stride, 4), torch.int32),
                           ctx.empty((B * plan.state_stride,), torch.uint8), ctx.empty((B, 3, 4), torch.int64))
                    ctx.check(ctx.lib.aej_test_fail_after_stage(ctx.handle, stage))
                    with pytest.raises(AejError, match="injected failure"):
                        codec.encode_into(ctx, x, plan, *out)
                    assert s.query(), f"{nsub} sub-batch(es), failure after {name}: the caller's stream still has work in flight"
                    # nothing of the failed call is in flight on the library's private streams either: the next call's outputs are right
                    with pytest.raises(AejError):
                        codec.encode_end(ctx)                      # and no call is left pending
                    got = _encode_raw(torch, ctx, codec, x, plan)
                    _assert_same_encoding(torch, plan, got, want, B, f"call after a failure behind {name} ({nsub} sub-batch(es))")
                    # begin / end flavour: a failing begin returns the error itself and leaves nothing to end
                    ctx.check(ctx.lib.aej_test_fail_after_stage(ctx.handle, stage))
                    with pytest.raises(AejError, match="injected failure"):
                        codec.encode_begin(ctx, x, plan, *out)
                    ctx._in_flight = None
                    with pytest.raises(AejError):
                        codec.encode_end(ctx)
            # a stand-alone Canny on the same context afterwards still works (no stale chain hook re-recording a shared event)
            plane = torch.rand((256, 320), device=dev).cpu().numpy().astype(np.float32)
            e1 = A.EdgeDetection.canny(plane)
            assert e1.shape == plane.shape
        finally:
            ctx.set_sub_batches(0)
            ctx.check(ctx.lib.aej_test_fail_after_stage(ctx.handle, -1))
    check_image(EncodedBatch(plan, *want), 0, oracle.encode_image(x[0].cpu().numpy(), space, qr, br), "reference call")


def test_decode_refuses_tables_that_leave_the_plan(env):
    """include/aej.h: aej_decode_batch returns AEJ_ERR_ARG for a leaf table whose entries would make a kernel touch memory the plan
    does not own -- origin outside the layer, negative or overlong coefficient offset, size outside the block range."""
    torch, A, bench = env
    dev = torch.device("cuda", 0)
    x = bench.synth_batch(torch, 2, 256, 384, 5, dev)
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    enc = codec.compress_batch(x)
    good = codec.decompress_batch(enc)
    p = enc.plan
    row = int(p.leaf_stride) + int(p.leaf_off[1]) + 3                  # a leaf of image 1, layer 1
    cap1 = int(p.coeff_off[2] - p.coeff_off[1])
    for col, val in ((0, -4), (0, int(p.layer_w[1])), (1, -8), (1, int(p.layer_h[1]) + 64), (3, -1), (3, cap1), (2, 3), (2, 128)):
        saved = enc.leaves[row, col].item()
        enc.leaves[row, col] = val
        with pytest.raises(ValueError):
            codec.decompress_batch(enc)
        enc.leaves[row, col] = saved
    assert torch.equal(codec.decompress_batch(enc), good)


def test_schedule_report_follows_the_stated_hardware_queues(env):
    """aej_set_hw_queues / aej_get_schedule_host: the library schedules for the queue count the host states and says when fewer than
    8 queues force the narrower schedule."""
    torch, A, bench = env
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    ctx = codec._bind()
    n, src = A.hw_queues()
    try:
        ctx.check(ctx.lib.aej_set_hw_queues(ctx.handle, 16))
        s16 = ctx.schedule(64, 2160, 3840)
        assert s16["hw_queues"] == 16 and s16["sub_batches"] == 4 and not s16["limited_by_hw_queues"]
        ctx.check(ctx.lib.aej_set_hw_queues(ctx.handle, 4))
        s4 = ctx.schedule(64, 2160, 3840)
        assert s4["hw_queues"] == 4 and s4["sub_batches"] <= 2 and s4["limited_by_hw_queues"]
        assert ctx.schedule(1, 1080, 1920)["sub_batches"] == 1
    finally:
        ctx.check(ctx.lib.aej_set_hw_queues(ctx.handle, n))


def test_both_64x64_dct_kernels_and_both_plane_layouts_match_the_oracle(env, oracle):
    """launch_dct picks the one-wave 64 x 64 kernel for calls that have the device to themselves and the four-wave kernel for sub-batched /
    pipelined ones (DctArgs::crowded); the normalised planes of strip-kernel shapes are kept in 4 x 4 blocks (Geom::tiled) unless the option
    planes_row_major says otherwise.  Every pairing of the two choices (aej_set_option "dct64_kernel" = 1 / 4, "planes_row_major" = 0 / 1) is
    compared with the ORACLE directly -- a batch large enough for the strip kernel's 32- / 64-row strips, planes that clip the last row of
    64 x 64 leaves, CLAHE tiles 4 (mod 8) rows high (chroma blocks split between half-waves), in two of the matrix colour spaces."""
    torch, A, bench = env
    B, H, W = 12, 1072, 1920                     # 1072 / 4 = 268 = 4 (mod 8)
    x = bench.synth_batch(torch, B, H, W, 11, torch.device("cuda", 0))
    picks = (0, 5, 11)
    for space in ("YCbCr", "YCoCg-R"):
        codec = A.Jpeg(A.JpegCompressionSettings(space, (40, 80), (4, 64)))
        refs = {b: oracle.encode_image(x[b].cpu().numpy(), space, (40, 80), (4, 64)) for b in picks}
        assert sum(int((refs[b][l]["leaves"][:, 2] == 64).sum()) for b in picks for l in range(3)) > 50, "the images must contain 64 x 64 leaves"
        ctx = codec._bind()
        try:
            for kernel in (1, 4):
                for row_major in (0, 1):
                    ctx.set_option("dct64_kernel", kernel)
                    ctx.set_option("planes_row_major", row_major)
                    assert ctx.get_option("dct64_kernel") == kernel and ctx.get_option("planes_row_major") == row_major
                    enc = codec.compress_batch(x)
                    for b in picks:
                        check_image(enc, b, refs[b], f"{space} dct64_kernel={kernel} planes_row_major={row_major} image {b}")
        finally:
            ctx.set_option("dct64_kernel", 0)
            ctx.set_option("planes_row_major", 0)
    with pytest.raises(ValueError):
        ctx.set_option("dct64_kernel", 2)
    with pytest.raises(ValueError):
        ctx.set_option("no_such_option", 1)


def test_merged_dct_launch_and_xcd_tile_order_change_no_result(env, oracle):
    """Round 5's launch shapes: calls of at most 8 Mpx run the DCTs of sizes 4 .. 64 as one launch (k_dct_multi, option "dct_multi"), and the Sobel /
    NMS kernel walks its tiles in XCD-contiguous runs (option "sobel_xcd").  Every pairing, and the merged launch on both plane layouts, against the
    ORACLE; a call above 8 Mpx (per-size launches whatever the option says) must agree with the small calls image by image."""
    torch, A, bench = env
    B, H, W = 3, 1080, 1920
    x = bench.synth_batch(torch, B, H, W, 23, torch.device("cuda", 0))
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    refs = {b: oracle.encode_image(x[b].cpu().numpy(), "YCbCr", (40, 80), (4, 64)) for b in (0, 2)}
    sizes = np.concatenate([refs[b][l]["leaves"][:, 2] for b in refs for l in range(3)])
    assert all((sizes == s).any() for s in (4, 8, 16, 32, 64)), "every block size must occur"
    ctx = codec._bind()
    try:
        for multi in (1, 0):
            for xcd in (1, 0):
                for row_major in ((0, 1) if multi else (0,)):
                    ctx.set_option("dct_multi", multi); ctx.set_option("sobel_xcd", xcd); ctx.set_option("planes_row_major", row_major)
                    enc = codec.compress_batch(x)
                    for b in refs:
                        check_image(enc, b, refs[b], f"dct_multi={multi} sobel_xcd={xcd} planes_row_major={row_major} image {b}")
        ctx.set_option("dct_multi", 1); ctx.set_option("sobel_xcd", 1); ctx.set_option("planes_row_major", 0)
        big = codec.compress_batch(torch.cat([x, x]))                      # 12.4 Mpx: above the merged launch's limit
        for b in refs:
            check_image(big, b + B, refs[b], f"call above 8 Mpx, image {b + B}")
    finally:
        ctx.set_option("dct_multi", 1); ctx.set_option("sobel_xcd", 1); ctx.set_option("planes_row_major", 0)

