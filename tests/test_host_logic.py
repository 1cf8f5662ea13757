"""CPU: the product's host-side logic (no GPU calls): tables, settings, error behaviour, C-ABI symbols."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest

import adaptive_edge_aware_jpeg_amd as A
from adaptive_edge_aware_jpeg_amd import tables
from conftest import GOLDEN, ROOT


def test_zigzag_quality_tables_match_reference_goldens():
    z = np.load(os.path.join(GOLDEN, "zigzag.npz"))
    for k in z.files:
        assert np.array_equal(tables.zigzag_ordering(int(k[1:])), z[k])
        assert np.array_equal(A.Jpeg._zigzag_ordering(int(k[1:])), z[k])
    q = json.load(open(os.path.join(GOLDEN, "quality.json")))
    for key, tab in q["quality"].items():
        br, qr = (tuple(int(v) for v in part.split("-")) for part in key.split("|"))
        for s, v in tab.items():
            assert tables.quality_factor(int(s), br, qr) == v
    for n, lp in q["largest_power_of_2"].items():
        assert tables.largest_power_of_2(int(n)) == lp
    assert sorted(A.get_color_spaces()) == q["color_spaces"]
    d = A.JpegCompressionSettings()
    assert [d.color_space, list(d.quality_range), list(d.block_size_range)] == [q["defaults"]["color_space"], q["defaults"]["quality_range"], q["defaults"]["block_size_range"]]
    for sp, r in q["ratios"].items():
        assert A.JpegCompressionSettings(sp).downsampling_ratios.tolist() == r


def test_quant_matrices_match_reference_built_ones():
    meta = json.load(open(os.path.join(GOLDEN, "compress_cases.json")))
    for name, m in meta.items():
        j = A.Jpeg(A.JpegCompressionSettings(m["space"], tuple(m["quality_range"]), tuple(m["block_size_range"])))
        for key, ref in m["qm"].items():
            l, s = (int(v) for v in key.split("_"))
            assert j.quantization_matrix_cache[l][s].tolist() == ref
            assert j.quantization_matrix_cache[l][s].dtype == np.int32


def test_quant_matrices_match_oracle_over_many_settings(oracle):
    for q in (1, 10, 20, 40, 49, 50, 51, 75, 99):
        for s in (2, 4, 8, 16, 32, 64, 128, 256):
            for T in (tables.LUMINANCE_QUANTIZATION_MATRIX, tables.CHROMINANCE_QUANTIZATION_MATRIX):
                assert np.array_equal(tables.quantization_matrix(T, s, q), oracle.quant_matrix(T, s, q))
                assert tables.quantization_matrix(T, s, q).min() >= 1
    assert np.array_equal(tables.quantization_matrix(tables.LUMINANCE_QUANTIZATION_MATRIX, 8, 50).astype(np.float32),
                          tables.LUMINANCE_QUANTIZATION_MATRIX)           # quality 50 leaves the table unchanged


def test_layer_shapes_and_decode_leaf_sizes():
    j = A.Jpeg(A.JpegCompressionSettings("ICtCp"))
    j.update_layer_shapes((1080, 1920))
    assert j.layer_shapes.tolist() == [[1080, 1920], [1080, 480], [1080, 480]]
    j = A.Jpeg(A.JpegCompressionSettings("YCbCr"))
    j.update_layer_shapes((2160, 3840))
    assert j.layer_shapes.tolist() == [[2160, 3840], [1080, 1920], [1080, 1920]]
    assert A.Jpeg._decode_leaf_sizes([1, 0, 1, 0, 0, 2, 0, 0, 2], 16) == [8, 4, 4, 4, 8]


def test_error_behaviour_matches_reference():
    with pytest.raises(ValueError, match="Unsupported color space"):
        A.JpegCompressionSettings("CMYK")
    j = A.Jpeg(A.JpegCompressionSettings())
    with pytest.raises(TypeError):
        j.compress(np.zeros((4, 4, 3), np.float32))                       # jpeg.py:250-251
    with pytest.raises(ValueError):
        j.compress(A.Image(np.zeros((4, 4), np.float32), (4, 4), None))   # jpeg.py:252-253
    with pytest.raises(TypeError):
        A.EdgeDetection.canny([[0.0]])
    with pytest.raises(ValueError):
        A.EdgeDetection.canny(np.zeros((2, 2, 2), np.float32))
    with pytest.raises(TypeError):
        A.QuadTree([[0]])
    with pytest.raises(ValueError):
        A.QuadTree(np.zeros((2, 2, 2), np.float32))
    with pytest.raises(TypeError):
        A.convert("sRGB", "YCbCr", [[0, 0, 0]])
    with pytest.raises(ValueError):
        A.convert("sRGB", "YCbCr", np.zeros((4, 2), np.float32))
    with pytest.raises(ValueError):
        A.convert("sRGB", "nope", np.zeros((4, 3), np.float32))
    with pytest.raises(ValueError):
        A.convert("YCbCr", "OKLAB", np.zeros((4, 3), np.float32))
    with pytest.raises(ValueError):
        A.apply_normalization("nope", np.zeros((4, 3), np.float32), False)


def test_apply_normalization_values():
    c = np.load(os.path.join(GOLDEN, "color_forward.npz"))
    for sp in ("YCbCr", "YCoCg", "YCoCg-R", "ICaCb"):
        assert np.array_equal(A.apply_normalization(sp, c[sp], False).astype(np.float32), c[sp + "_norm"])
        back = A.apply_normalization(sp, c[sp + "_norm"], True)
        assert np.abs(back - c[sp]).max() < 1e-5


def test_image_container(tmp_path, lena):
    img = A.Image.load(os.path.join(GOLDEN, "lena.png"))
    assert img.data.dtype == np.float32 and img.data.shape == (512, 512, 3) and img.extension == ".png"
    assert np.array_equal(img.data, lena)
    assert img.get_flattened().shape == (512 * 512, 3)
    cp = img.copy()
    cp.data[0, 0, 0] = 0.5
    assert img.data[0, 0, 0] != 0.5
    p = tmp_path / "o.png"
    img.save(str(p))
    assert np.array_equal(A.Image.load(str(p)).get_uint8(), img.get_uint8())
    g = A.Image.from_array(np.zeros(12, np.float32), (2, 2, 3))
    assert g.data.shape == (2, 2, 3)


def test_cabi_library_exports_every_declared_symbol():
    """No compute: the library loads and every function declared in include/*.h (the boundary aej.h and the test-only aej_testing.h) resolves."""
    from adaptive_edge_aware_jpeg_amd import _lib
    header = open(os.path.join(ROOT, "include", "aej.h")).read() + open(os.path.join(ROOT, "include", "aej_testing.h")).read()
    declared = set(re.findall(r"AEJ_API[^;(]*?\b(aej_\w+)\s*\(", header))
    assert len(declared) >= 20
    lib = _lib.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.aej_abi_version() == 3
    # host-only geometry helpers (no device needed)
    lc, sc, cc = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    assert lib.aej_quadtree_capacity(2160, 3840, 4, 64, ctypes.byref(lc), ctypes.byref(sc), ctypes.byref(cc)) == 0
    assert cc.value == 3840 * 2176 and lc.value == (3840 // 4) * (2176 // 4)
    assert lib.aej_canny_workspace_bytes(1080, 1920) > 2 * 1080 * 1920
    assert lib.aej_quadtree_workspace_bytes(1080, 1920, 4, 64) > 0
    assert lib.aej_quadtree_capacity(100, 100, 3, 64, None, None, None) != 0       # not a power of two
    assert b"context" in lib.aej_last_error(None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "adaptive_edge_aware_jpeg_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the CPU oracle", "").replace("CPU oracle", ""), os.path.join(dirpath, f)


def test_leaf_positions_host_helper(oracle):
    """aej_leaf_positions_host (host code of the library, Jpeg._block_merge's walk) against the oracle and the encoder's own leaves."""
    from adaptive_edge_aware_jpeg_amd import _lib
    lib = _lib.load_library()
    rng = np.random.default_rng(3)
    for (h, w, mn, mx) in ((150, 211, 4, 64), (64, 64, 8, 8), (300, 77, 4, 128), (33, 50, 2, 16)):
        edge = (rng.random((h, w)) < 0.01).astype(np.uint8)
        leaves, states, root = oracle.quadtree(edge, mn, mx)
        sizes = np.ascontiguousarray(leaves[:, 2], dtype=np.int32)
        xy = np.zeros((len(sizes), 2), np.int32)
        n = lib.aej_leaf_positions_host(sizes.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(sizes)), root, h, w,
                                        xy.ctypes.data_as(ctypes.c_void_p))
        assert n == len(sizes) and np.array_equal(xy, leaves[:, :2])
        assert np.array_equal(oracle.leaf_positions(sizes, root, h, w), leaves[:, :2])


def test_compat_alias_packages_expose_the_reference_import_names():
    """`from jpeg import ...`, `from jpeg.utils import largest_power_of_2`, `from jpeg.quadtree import QuadTree`,
    `from color import convert`, `from image import Image` (setup.py:26-27; src/jpeg/utils.py:24-41) resolve with compat/ on
    the path.  Run in a child interpreter so the alias names do not leak into this test process."""
    import subprocess
    code = (
        "import sys; sys.path.insert(0, sys.argv[1])\n"
        "from jpeg import Jpeg, JpegCompressionSettings\n"
        "from jpeg.utils import largest_power_of_2\n"
        "from jpeg.jpeg import Jpeg as J2\n"
        "from jpeg.quadtree import QuadTree, QuadNode\n"
        "from jpeg.edge_detection import EdgeDetection\n"
        "from color import convert, apply_normalization, get_color_spaces\n"
        "from image import Image, EvaluationMetrics\n"
        "assert J2 is Jpeg\n"
        "assert [largest_power_of_2(n) for n in (1, 2, 3, 512, 513, 1920, 3840, 7680)] == [1, 2, 2, 256, 512, 1024, 2048, 4096]\n"
        "try:\n    largest_power_of_2(0)\nexcept ValueError:\n    pass\nelse:\n    raise SystemExit('no ValueError')\n"
        "assert sorted(get_color_spaces()) == sorted(['ICaCb', 'ICtCp', 'JzAzBz', 'OKLAB', 'YCbCr', 'YCoCg', 'YCoCg-R'])\n"
        "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "compat")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_hardware_queue_policy_import_has_no_side_effect():
    """_lib: importing the package never writes the environment; the library is told a queue count only
    when the package can be sure the HIP runtime starts (or started) with it -- variable already in the environment, or the explicit
    configure_hw_queues() before HIP is up; otherwise HIP's default of 4."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")

    def run(code, extra=None):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, **(extra or {})), timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        return r.stdout.strip().splitlines()[-1]

    probe = "import adaptive_edge_aware_jpeg_amd as A, os; print(A.hw_queues()[0], os.environ.get('GPU_MAX_HW_QUEUES'))"
    assert run(probe) == "4 None"                                                   # import: environment untouched, HIP's default assumed
    assert run("import torch\n" + probe) == "4 None"
    assert run("import torch\n" + probe, {"GPU_MAX_HW_QUEUES": "12"}) == "12 12"   # present at process start: trusted as it is
    opt_in = "import adaptive_edge_aware_jpeg_amd as A, os; A.configure_hw_queues(); print(A.hw_queues()[0], os.environ.get('GPU_MAX_HW_QUEUES'))"
    assert run(opt_in) == "16 16"                                                   # explicit opt-in before torch: set and trusted
    assert run(opt_in, {"GPU_MAX_HW_QUEUES": "8"}) == "8 8"                         # never overrides the launcher
    # after `import torch` nobody can know whether the runtime has read the variable (device_count() / is_available() initialise HIP):
    # the variable is set, the library keeps HIP's default
    assert run("import torch\n" + opt_in) == "4 16"
    assert run("import adaptive_edge_aware_jpeg_amd as A; A.set_hw_queues(24); print(A.hw_queues()[0])") == "24"


def test_deflate_tables_round_trip_through_zlib():
    """Host side of the opt-in GPU entropy stage: the Huffman tables / dynamic-block headers (restated in tests/deflate_reference.py, the
    construction aej_deflate_build_tables follows), driven through a small token-level encoder, must give streams `zlib.decompress` (the
    reference's decoder call, jpeg.py:659) reads back exactly -- fixed code, adaptive code, a table counted on OTHER data, empty / tiny /
    incompressible inputs, matches at every distance class."""
    import zlib
    import deflate_reference as DT
    rng = np.random.default_rng(0)

    def coeff_like(n, p_nonzero, scale):
        v = np.zeros(n, np.int32)
        m = rng.random(n) < p_nonzero
        v[m] = np.round(rng.laplace(0, scale, int(m.sum()))).astype(np.int32)
        return v.tobytes()
    cases = {"sparse": coeff_like(1500, 0.05, 2), "dense": coeff_like(1200, 0.6, 6), "large values": coeff_like(900, 0.3, 3000),
             "zeros": bytes(3000), "one coefficient": b"\x01\x00\x00\x00", "empty": b"", "noise": rng.integers(0, 256, 1500, dtype=np.uint8).tobytes()}
    fixed = DT.fixed_table()
    assert fixed.shape == (DT.TABLE_WORDS,) and int(fixed[DT.HDR_BITS_AT]) == 3
    foreign = None
    for name, data in cases.items():
        tok = DT.greedy_tokens(data)
        hist = DT.histogram_of(tok)
        assert zlib.decompress(DT.encode_tokens(data, tok, fixed)) == data, name
        own = DT.adaptive_table(hist[:286], hist[286:316])
        assert int(own[DT.HDR_BITS_AT]) <= DT.HEADER_WORDS * 32 and all(0 < (int(e) >> 16) <= 15 for e in own[:316])
        assert zlib.decompress(DT.encode_tokens(data, tok, own)) == data, name
        if foreign is None:
            foreign = own                       # (a cover-everything table counted on the first case: valid for every other one)
        assert zlib.decompress(DT.encode_tokens(data, tok, foreign)) == data, name + " (foreign table)"
        # exact tables (cover_all=False: what Jpeg.deflate_batch builds -- codes only for the symbols the SAME parse contains): valid, and never
        # larger than the cover-everything table
        exact = DT.adaptive_table(hist[:286], hist[286:316], cover_all=False)
        enc_exact = DT.encode_tokens(data, tok, exact)
        assert zlib.decompress(enc_exact) == data, name + " (exact table)"
        assert len(enc_exact) <= len(DT.encode_tokens(data, tok, own)) + 16, name
        if name in ("sparse", "zeros", "one coefficient", "empty"):
            assert int(exact[DT.HDR_BITS_AT]) < int(own[DT.HDR_BITS_AT]) - 100, name      # few symbols: a shorter block header
    # every distance class and the longest length, from a hand-made parse: 40 random bytes, a long zero run, the 40 bytes again from 32 000 back
    far = bytes(rng.integers(1, 256, 40, dtype=np.uint8))
    tok = [("lit", b) for b in far] + [("lit", 0)] + [("match", 258, 1)] * 123 + [("match", 225, 1)] + [("match", 40, 32000)]
    data = far + bytes(32000 - 40) + far
    for d in (1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 32040):
        tok.append(("match", 3, d))
        data += data[len(data) - d:len(data) - d + 3] if d >= 3 else (data[-d:] * 3)[:3]
    hist = DT.histogram_of(tok)
    assert (hist[286:316] > 0).all()
    assert zlib.decompress(DT.encode_tokens(data, tok, DT.adaptive_table(hist[:286], hist[286:316], cover_all=False))) == data
    assert zlib.decompress(DT.encode_tokens(data, tok, fixed)) == data
    # length-limited Huffman: Kraft equality for a skewed histogram that plain Huffman would give codes longer than 15 bits
    skew = [2 ** i for i in range(30)]
    ls = DT.huffman_lengths(skew, 15)
    assert max(ls) <= 15 and sum(2.0 ** -l for l in ls) <= 1.0 + 1e-12


def test_bench_asks_for_more_hardware_queues_when_it_will_hold_a_process_group(monkeypatch):
    """A live RCCL communicator creates streams of its own; with the 16 queues a lone process uses it pushed the library's streams onto
    shared queues (13 % of the step, profiles/r04_rccl_hw_queues.txt): a process that will hold a process group asks for 24 (and creates
    the communicator after its own streams -- bench.init_distributed passes no device_id)."""
    import importlib
    import inspect
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    monkeypatch.delenv("AEJ_BENCH_FORCE_DIST", raising=False)
    assert bench.hw_queue_default(1) == "16" and bench.hw_queue_default(2) == "24" and bench.hw_queue_default(8) == "24"
    monkeypatch.setenv("AEJ_BENCH_FORCE_DIST", "1")
    assert bench.hw_queue_default(1) == "24"
    src = inspect.getsource(bench.init_distributed)
    eager = src.index('AEJ_BENCH_NCCL_EAGER')
    assert "device_id" in src[eager:src.index("else:", eager)] and "device_id" not in src[src.index("else:", eager):]      # only the A / B switch binds the device early


def test_library_builds_the_same_huffman_tables_as_the_python_restatement():
    """aej_deflate_build_tables (host-only C++ in csrc/deflate.hip, what Jpeg.deflate_batch calls) against tests/deflate_reference.py's
    adaptive_table (the readable restatement): word for word, for skewed, flat, sparse and empty histograms, in both cover modes --
    including histograms whose plain Huffman code would be longer than 15 bits."""
    import deflate_reference as DT
    from adaptive_edge_aware_jpeg_amd import deflate_tables as PKG
    from adaptive_edge_aware_jpeg_amd._lib import load_library
    assert (PKG.HIST_BINS, PKG.TABLE_WORDS) == (DT.HIST_BINS, DT.TABLE_WORDS)
    header = open(os.path.join(ROOT, "include", "aej.h")).read()
    assert f"#define AEJ_DEFLATE_HIST_BINS {DT.HIST_BINS}" in header and f"#define AEJ_DEFLATE_TABLE_WORDS {DT.TABLE_WORDS}" in header
    lib = load_library()
    rng = np.random.default_rng(7)
    nb = DT.HIST_BINS
    hists = [np.zeros(nb, np.int32), np.ones(nb, np.int32)]
    for trial in range(60):
        h = (rng.random(nb) ** int(rng.integers(1, 8)) * 10.0 ** int(rng.integers(1, 8))).astype(np.int64)
        h[rng.random(nb) < rng.random()] = 0                        # a random share of symbols never occurs
        if trial % 5 == 0:
            h[286:316] = 0                                          # a stream without a single match
        hists.append(np.minimum(h, 2 ** 31 - 1).astype(np.int32))
    fib = np.zeros(nb, np.int32); fib[:40] = [min(int(1.6 ** k), 2 ** 30) for k in range(40)]      # plain Huffman: depths far beyond 15
    fib[286:316] = fib[:30]
    hists.append(fib)
    for k in range(0, len(hists) - 2, 3):
        hist = np.ascontiguousarray(np.stack(hists[k:k + 3]), dtype=np.int32)
        for cover in ([1, 1, 1], [0, 0, 0], [1, 0, 1]):
            cov = np.array(cover, np.int32)
            tab = np.full((3, DT.TABLE_WORDS), 0xdeadbeef, np.uint32)
            assert lib.aej_deflate_build_tables(hist.ctypes.data, cov.ctypes.data, tab.ctypes.data) == 0
            for l in range(3):
                want = DT.adaptive_table(hist[l, :286], hist[l, 286:316], cover_all=bool(cover[l]))
                assert np.array_equal(tab[l], want), (k, l, cover, np.flatnonzero(tab[l] != want)[:8])
    # NULL cover = every symbol gets a code
    tab = np.zeros((3, DT.TABLE_WORDS), np.uint32)
    hist = np.ascontiguousarray(np.stack(hists[2:5]), dtype=np.int32)
    assert lib.aej_deflate_build_tables(hist.ctypes.data, None, tab.ctypes.data) == 0
    assert all(np.array_equal(tab[l], DT.adaptive_table(hist[l, :286], hist[l, 286:316])) for l in range(3))


def test_pack_u8_levels_host_helper():
    """aej_pack_u8_levels_host (include/aej.h): a host float32 batch is taken for 8-bit ingest only when EVERY value is bit for bit
    float32(k) / 255 -- what Image.load produces (src/image/image.py:80) -- and the levels it writes are those k."""
    from adaptive_edge_aware_jpeg_amd import _lib
    lib = _lib.load_library()
    rng = np.random.default_rng(5)
    u = rng.integers(0, 256, size=(70, 333, 3), dtype=np.uint8)
    u.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)                      # every level occurs
    f = u.astype(np.float32) / np.float32(255.0)
    out = np.empty(f.shape, np.uint8)
    for threads in (1, 3, 64):
        out[:] = 0
        assert lib.aej_pack_u8_levels_host(f.ctypes.data, f.size, out.ctypes.data, threads) == 1
        assert np.array_equal(out, u)
    for bad in (np.nextafter(f[3, 7, 1], np.float32(2)), np.float32(np.nan), np.float32(-0.0), np.float32(1.5), np.float32(-1.0), np.float32(np.inf),
                np.float32(0.5)):                                             # 0.5 = 127.5 / 255 is no level
        g = f.copy()
        g[3, 7, 1] = bad
        assert lib.aej_pack_u8_levels_host(g.ctypes.data, g.size, out.ctypes.data, 2) == 0, bad
    r = rng.random((64, 64, 3), dtype=np.float32)
    assert lib.aej_pack_u8_levels_host(r.ctypes.data, r.size, out.ctypes.data, 2) == 0
    assert lib.aej_pack_u8_levels_host(f.ctypes.data, 0, out.ctypes.data, 2) == 1      # an empty batch is trivially all levels
    assert lib.aej_pack_u8_levels_host(None, 4, out.ctypes.data, 2) < 0


def test_bench_sub_batch_policy_for_pipelined_steps():
    """bench.py asks for two sub-batches per call only where it keeps four or more 64 x 4K-sized calls in flight; everywhere else (and for the
    blocking calls always) the library's automatic choice or the explicit --sub-batches stands (profiles/r05_sched_sweep.txt)."""
    import types
    import bench
    a = types.SimpleNamespace(pipelined_sub_batches=-1, sub_batches=0)
    assert bench.pipelined_sub_batches(a, 4, 64, 2160, 3840) == 2
    assert bench.pipelined_sub_batches(a, 6, 64, 2160, 3840) == 2
    assert bench.pipelined_sub_batches(a, 3, 64, 2160, 3840) == 0          # three contexts: automatic (four)
    assert bench.pipelined_sub_batches(a, 4, 64, 1080, 1920) == 0          # smaller calls: automatic
    assert bench.pipelined_sub_batches(a, 4, 8, 4320, 7680) == 0
    a.sub_batches = 4
    assert bench.pipelined_sub_batches(a, 4, 64, 2160, 3840) == 4          # an explicit --sub-batches applies to both kinds of call
    a.pipelined_sub_batches = 3
    assert bench.pipelined_sub_batches(a, 4, 64, 2160, 3840) == 3
    assert bench.parse_args(["--gpus", "1"]).pipeline == 4 and bench.parse_args([]).steps == 20

