import os, sys, zlib, numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
import adaptive_edge_aware_jpeg_amd as A
from adaptive_edge_aware_jpeg_amd import deflate_tables as DT
dev = torch.device("cuda", 0)
x = bench.synth_batch(torch, 1, 768, 1024, 5, dev)
codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
enc = codec.compress_batch(x)
ctx = codec._bind(); t = ctx.torch; p = enc.plan
hist = ctx.empty((3, 288), t.int32)
ctx.check(ctx.lib.aej_deflate_histogram(ctx.handle, enc.coeffs.data_ptr(), enc.counts.data_ptr(), p.batch, p.H, p.W, hist.data_ptr()))
h = hist.cpu().numpy()
for l in range(3):
    raw = enc.layer(0, l)["coeffs"].tobytes()
    lit, dist = DT.histogram_reference(raw)
    print("layer", l, len(raw), "hist equal:", np.array_equal(h[l, :286], lit), np.array_equal(h[l, 286:288], dist), "tokens", int(lit.sum()), int(h[l,:286].sum()))
    T = DT.adaptive_table(lit, dist)
    ref = DT.encode_reference(raw, T)
    g = codec.deflate_batch(enc, tables=np.stack([T] * 3))[0][l]
    gf = codec.deflate_batch(enc, adaptive=False)[0][l]
    print("   sizes: gpu(table)", len(g), "python(all dynamic)", len(ref), "gpu fixed", len(gf), "python fixed", len(DT.encode_reference(raw, DT.fixed_table())), "zlib9", len(zlib.compress(raw, 9)))
