# per-kernel times of the opt-in GPU entropy stage on natural 4K images (developer measurement)
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
import adaptive_edge_aware_jpeg_amd as A
dev = torch.device("cuda", 0)
x = bench.natural_batch(torch, 8, 2160, 3840, 3, dev)
codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
enc = codec.compress_batch(x)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    s = codec.deflate_batch(enc, adaptive=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("deflate_batch %.1f ms, %d bytes" % ((t1 - t0) * 1e3, sum(len(b) for im in s for b in im)))
t0 = time.perf_counter(); out = codec.compress_many(x, entropy="gpu"); t1 = time.perf_counter()
print("compress_many gpu %.1f ms" % ((t1 - t0) * 1e3))
# where compress_many(entropy="gpu") spends its time
import numpy as np
def section(name, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); print("  %-28s %.2f ms" % (name, (t1 - t0) * 1e3)); return t1
for rep in range(2):
    t = time.perf_counter()
    enc = codec.compress_batch(x); t = section("compress_batch", t)
    p = enc.plan
    streams = codec.deflate_batch(enc, adaptive=True); t = section("deflate_batch", t)
    cnt = enc.counts_host; t = section("counts_host", t)
    sts = [[enc.states[b * p.state_stride + p.state_off[l]: b * p.state_stride + p.state_off[l] + int(cnt[b, l, 2])].cpu().numpy() for l in range(3)] for b in range(p.batch)]
    t = section("24 state copies", t)
    recs = [[codec._layer_bytes({"states": sts[b][l], "root_size": int(cnt[b, l, 3])}, stream=streams[b][l]) for l in range(3)] for b in range(p.batch)]
    t = section("24 x _layer_bytes", t)
    print("  states bytes", sum(len(s) for im in sts for s in im))
