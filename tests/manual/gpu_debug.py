"""Ad-hoc stage-by-stage GPU-vs-oracle report (not a pytest file). Run on the GPU box:
    python tests/manual/gpu_debug.py > gpurun_out/debug.log 2>&1
"""
import ctypes, os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import adaptive_edge_aware_jpeg_amd as A
from adaptive_edge_aware_jpeg_amd._lib import get_context, SPACE_IDS
from oracle import oracle as O

def report(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape:
        print(f"[FAIL] {name}: shape {a.shape} vs {b.shape}"); return False
    if a.dtype.kind == 'f':
        same = (a == b) | (np.isnan(a) & np.isnan(b))
    else:
        same = a == b
    nbad = int((~same).sum())
    if nbad == 0:
        print(f"[ ok ] {name}: {a.size} values identical"); return True
    idx = np.argwhere(~same)[:5]
    print(f"[FAIL] {name}: {nbad}/{a.size} differ; first {idx.tolist()} gpu={[a[tuple(i)] for i in idx]} ora={[b[tuple(i)] for i in idx]}")
    return False

def step(fn):
    try:
        fn()
    except Exception:
        traceback.print_exc()

def t_color():
    rng = np.random.default_rng(3)
    x = (rng.integers(0, 256, size=(20000, 3)).astype(np.float32) / np.float32(255))
    for sp in SPACE_IDS:
        y = A.convert("sRGB", sp, x)
        report(f"convert {sp}", y, O.color_forward(sp, x))

def t_planes():
    ctx = get_context()
    for sp, (H, W) in (("YCbCr", (64, 128)), ("ICtCp", (34, 72)), ("OKLAB", (18, 36))):
        img = O.synth_image(H, W, 5).astype(np.float32) / np.float32(255)
        j = A.Jpeg(A.JpegCompressionSettings(sp)); c = j._bind()
        plan = c.plan(1, H, W)
        n = sum(((plan.layer_h[l] * plan.layer_w[l] + 63) // 64) * 64 for l in range(3))
        raw = c.empty((n,), torch.float32); nrm = c.empty((n,), torch.float32); u8 = c.empty((n,), torch.uint8)
        x = c.to_device(img[None], torch.float32)
        c.check(c.lib.aej_color_planes(c.handle, x.data_ptr(), 1, H, W, raw.data_ptr(), nrm.data_ptr(), u8.data_ptr()))
        raw = raw.cpu().numpy(); nrm = nrm.cpu().numpy(); u8 = u8.cpu().numpy()
        conv = O.color_forward(sp, img.reshape(-1, 3)).reshape(H, W, 3)
        off = 0
        for l, (rh, rw) in enumerate(O.RATIOS[sp]):
            pl = O.downsample(conv, l, rh, rw); m = pl.size
            report(f"planes {sp} L{l} raw", raw[off:off + m].reshape(pl.shape), pl)
            report(f"planes {sp} L{l} norm", nrm[off:off + m].reshape(pl.shape), O.normalize(pl, sp, l))
            report(f"planes {sp} L{l} u8", u8[off:off + m].reshape(pl.shape), O.to_u8(pl))
            off += ((m + 63) // 64) * 64

def t_canny():
    for (H, W, seed) in ((96, 160, 1), (67, 101, 2), (270, 480, 3), (33, 50, 4), (512, 512, 5)):
        img = O.synth_image(H, W, seed).astype(np.float32) / np.float32(255)
        plane = O.color_forward("YCbCr", img.reshape(-1, 3)).reshape(H, W, 3)[:, :, 0].copy()
        if seed == 2:
            plane = plane - 0.5    # negative values: uint8 wrap
        e, st, thr = A.EdgeDetection.canny(plane, return_stages=True)
        eo, so, tho = O.edge_pipeline(plane, return_stages=True)
        for i, nm in enumerate(("scaled", "clahe", "gauss", "bilateral")):
            report(f"canny {H}x{W} {nm}", st[i], so[i])
        lo, hi = O.canny_thresholds(*tho)
        print(f"       thresholds gpu={thr} oracle={(lo, hi)} pct={tho}")
        _, nms = O.canny(so[3], tho[0], tho[1], return_nms=True)
        report(f"canny {H}x{W} nms", st[4], nms)
        report(f"canny {H}x{W} edges", e.astype(np.uint8), eo)
        print("       edge px:", int(eo.sum()))

def t_quadtree():
    g = np.load(os.path.join(ROOT, "tests/golden/quadtree_cases.npz"))
    n = int(g["n_cases"][0]); bad = 0
    for i in range(n):
        edge = g[f"c{i}_edge"].astype(np.float32); mn, mx, root = (int(v) for v in g[f"c{i}_params"])
        try:
            qt = A.QuadTree(edge, max_size=mx, min_size=mn)
        except NotImplementedError as ex:
            print(f"       case {i} {edge.shape} min{mn} max{mx}: unsupported ({ex})"); continue
        ok = np.array_equal(qt._leaves[:, :3], g[f"c{i}_leaves"]) and np.array_equal(qt._states, g[f"c{i}_states"])
        if not ok:
            bad += 1
            if bad < 6:
                print(f"[FAIL] quadtree case {i} shape {edge.shape} min{mn} max{mx} root{root}: leaves {qt._leaves.shape} vs {g[f'c{i}_leaves'].shape}, states {qt._states.shape} vs {g[f'c{i}_states'].shape}")
                print("       gpu states", qt._states[:24].tolist(), "ref", g[f"c{i}_states"][:24].tolist())
                print("       gpu leaves", qt._leaves[:6].tolist(), "ref", g[f"c{i}_leaves"][:6].tolist())
    print(f"[{'ok' if bad == 0 else 'FAIL'}] quadtree golden cases: {n - bad}/{n}")

def t_dct():
    ctx = get_context()
    rng = np.random.default_rng(9)
    for (bmin, bmax, H, W) in ((4, 64, 150, 210), (4, 128, 260, 300), (8, 8, 64, 64), (2, 16, 40, 56)):
        sp = "YCbCr"
        j = A.Jpeg(A.JpegCompressionSettings(sp, (40, 80), (bmin, bmax))); c = j._bind()
        norm = (rng.random((H, W), dtype=np.float32) * 254 - 127).astype(np.float32)
        edge = (rng.random((H, W)) < 0.004).astype(np.uint8)
        leaves, states, root = O.quadtree(edge, bmin, bmax)
        _, zz, qm = O.tables(sp, (40, 80), (bmin, bmax))
        for layer in (0, 1):
            co, do = O.blocks_encode(norm, leaves, qm[layer], zz, want_dct=True)
            offs = np.concatenate([[0], np.cumsum(leaves[:, 2].astype(np.int64) ** 2)[:-1]]).astype(np.int32)
            lv4 = np.concatenate([leaves, offs[:, None]], 1).astype(np.int32)
            d_l = c.to_device(lv4, torch.int32); d_n = c.to_device(norm, torch.float32)
            d_c = c.empty((co.size,), torch.int32); d_d = c.empty((co.size,), torch.float32)
            c.check(c.lib.aej_dct_quant_zigzag(c.handle, d_n.data_ptr(), H, W, layer, d_l.data_ptr(), ctypes.c_int64(len(lv4)), d_c.data_ptr(), d_d.data_ptr()))
            sizes = {int(s): int((leaves[:, 2] == s).sum()) for s in np.unique(leaves[:, 2])}
            report(f"dct {bmin}-{bmax} {H}x{W} L{layer} float (sizes {sizes})", d_d.cpu().numpy(), do)
            report(f"dct {bmin}-{bmax} {H}x{W} L{layer} coeffs", d_c.cpu().numpy(), co)

def t_full():
    from PIL import Image as PI
    lena = np.asarray(PI.open(os.path.join(ROOT, "tests/golden/lena.png")).convert("RGB")).astype(np.float32) / np.float32(255)
    cases = [("lena YCbCr 8-8 q50", lena, "YCbCr", (50, 50), (8, 8)),
             ("lena YCoCg 4-64", lena, "YCoCg", (40, 80), (4, 64)),
             ("synth 360x640 YCbCr", O.synth_image(360, 640, 20250718).astype(np.float32) / np.float32(255), "YCbCr", (40, 80), (4, 64)),
             ("synth 250x332 OKLAB 4-128", O.synth_image(250, 332, 11).astype(np.float32) / np.float32(255), "OKLAB", (40, 80), (4, 128)),
             ("synth 128x256 ICtCp", O.synth_image(128, 256, 12).astype(np.float32) / np.float32(255), "ICtCp", (20, 60), (4, 32))]
    for name, img, sp, qr, br in cases:
        j = A.Jpeg(A.JpegCompressionSettings(sp, qr, br))
        t0 = time.time(); enc = j.compress_batch(img[None], want_dct=True); torch.cuda.synchronize(); t1 = time.time()
        ora = O.encode_image(img, sp, qr, br)
        for l in range(3):
            L = enc.layer(0, l)
            ok = report(f"{name} L{l} states", L["states"], ora[l]["states"])
            ok &= report(f"{name} L{l} leaves", L["leaves"], ora[l]["leaves"])
            if ok:
                report(f"{name} L{l} coeffs", L["coeffs"], ora[l]["coeffs"])
            print(f"       root gpu={L['root_size']} ora={ora[l]['root_size']} leaves={len(L['leaves'])}")
        print(f"       gpu time {1e3*(t1-t0):.1f} ms")

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    for f in (t_color, t_planes, t_canny, t_quadtree, t_dct, t_full):
        print("=====", f.__name__); step(f)
