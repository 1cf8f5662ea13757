"""GPU entropy stage on natural 4K images (developer measurement, GPU box): every stream through zlib.decompress, bits per pixel against
zlib levels 9 / 6 / 1, per-call times.   python3 tools/profiling/deflate_check.py [images]"""
import os, sys, time, zlib
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import bench
import adaptive_edge_aware_jpeg_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
x = bench.natural_batch(torch, n, 2160, 3840, 777, dev)
codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
enc = codec.compress_batch(x)
torch.cuda.synchronize()
px = n * 2160 * 3840
for adaptive in (True, False):
    for rep in range(2):
        t0 = time.perf_counter()
        s = codec.deflate_batch(enc, adaptive=adaptive)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    nb = sum(len(b) for im in s for b in im)
    print(f"deflate_batch adaptive={adaptive}: {dt * 1e3:.1f} ms for {n} images = {px / dt / 1e6:.0f} MP/s, {nb} bytes = {8 * nb / px:.3f} bit/px")
    if adaptive:
        gpu_streams = s
ok = True
tot = {9: 0, 6: 0, 1: 0}
raw_bytes = 0
for b in range(n):
    for l in range(3):
        raw = enc.layer(b, l)["coeffs"].tobytes()
        raw_bytes += len(raw)
        ok = ok and zlib.decompress(gpu_streams[b][l]) == raw
        if b < 2:
            for lvl in tot:
                tot[lvl] += len(zlib.compress(raw, lvl))
print("every stream decompresses to its coefficients:", ok, f"({raw_bytes / 1e6:.0f} MB of coefficients)")
px2 = min(n, 2) * 2160 * 3840
g2 = sum(len(gpu_streams[b][l]) for b in range(min(n, 2)) for l in range(3))
print("first two images: gpu %.3f bit/px; zlib9 %.3f, zlib6 %.3f, zlib1 %.3f  -> gpu / zlib9 = %.3f" % (8 * g2 / px2, 8 * tot[9] / px2, 8 * tot[6] / px2, 8 * tot[1] / px2, g2 / tot[9]))
for rep in range(3):
    t0 = time.perf_counter(); out = codec.compress_many(x, extension=".png", entropy="gpu"); dt = time.perf_counter() - t0
    print("compress_many(entropy='gpu'): %.1f ms = %.0f MP/s, %.3f bit/px" % (dt * 1e3, px / dt / 1e6, 8 * sum(len(o) for o in out) / px))
# where compress_many(entropy="gpu") spends its time (host-side sections; each ends with a device synchronisation)
def section(name, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); print("  %-34s %.2f ms" % (name, (t1 - t0) * 1e3)); return t1
for rep in range(2):
    t = time.perf_counter()
    enc = codec.compress_batch(x); t = section("compress_batch", t)
    streams = codec.deflate_batch(enc, adaptive=True, as_views=True); t = section("deflate_batch", t)
    cnt = enc.counts_host; states = enc.states.cpu().numpy(); t = section("counts + states to the host", t)
    p = enc.plan
    out = [b"".join([codec._header_bytes(3)] + [piece for l in range(3) for piece in codec._layer_pieces({"states": states[b * p.state_stride + p.state_off[l]: b * p.state_stride + p.state_off[l] + int(cnt[b, l, 2])], "root_size": int(cnt[b, l, 3])}, stream=streams[b][l])]) for b in range(n)]
    t = section("container assembly", t)
