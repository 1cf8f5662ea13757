"""diagnostic (GPU box): area share of each leaf size on the bench workload"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import bench, adaptive_edge_aware_jpeg_amd as A
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = bench.synth_batch(torch, B, 2160, 3840, 20250718, torch.device("cuda", 0))
jpeg = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)), device=0)
ctx = jpeg._bind()
plan = ctx.plan(B, 2160, 3840)
coeffs = ctx.empty((B * plan.coeff_stride,), torch.int32)
leaves = ctx.empty((B * plan.leaf_stride, 4), torch.int32)
states = ctx.empty((B * plan.state_stride,), torch.uint8)
counts = ctx.empty((B, 3, 4), torch.int64)
jpeg.encode_into(ctx, x, plan, coeffs, leaves, states, counts)
torch.cuda.synchronize()
cnt = counts.cpu().numpy()
lv = leaves.cpu().numpy().reshape(B, plan.leaf_stride, 4)
tot = {}
for b in range(B):
    for l in range(3):
        lo, n = int(plan.leaf_off[l]), int(cnt[b, l, 1])
        s = lv[b, lo:lo + n, 2]
        for k in (4, 8, 16, 32, 64, 128):
            tot[k] = tot.get(k, 0) + int((s == k).sum()) * k * k
area = sum(tot.values())
print("leaf area share:", {k: round(v / area, 4) for k, v in sorted(tot.items())}, "px per image", area / B)
