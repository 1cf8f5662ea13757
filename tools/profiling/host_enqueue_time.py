"""diagnostic (GPU box): host time spent enqueuing (aej_encode_batch_begin) and waiting (aej_encode_batch_end) per pipelined step"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, adaptive_edge_aware_jpeg_amd as A
B, H, W = 64, 2160, 3840
dev = torch.device("cuda", 0)
xs = [bench.synth_batch(torch, B, H, W, s, dev) for s in (1, 2)]
jpeg = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)), device=0)
pipes = []
for i in range(2):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        ctx = jpeg._bind(); plan = ctx.plan(B, H, W)
        out = (ctx.empty((B * plan.coeff_stride,), torch.int32), ctx.empty((B * plan.leaf_stride, 4), torch.int32), ctx.empty((B * plan.state_stride,), torch.uint8), ctx.empty((B, 3, 4), torch.int64))
    pipes.append([ctx, plan, out, s, False])
tb = te = 0.0
def step(i, acc):
    global tb, te
    p = pipes[i % 2]
    with torch.cuda.stream(p[3]):
        if p[4]:
            t0 = time.perf_counter(); jpeg.encode_end(p[0]); te += (time.perf_counter() - t0) * acc
        t0 = time.perf_counter(); jpeg.encode_begin(p[0], xs[(i // 2 + i) & 1], p[1], *p[2]); tb += (time.perf_counter() - t0) * acc
        p[4] = True
for i in range(8): step(i, 0)
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for i in range(K): step(i, 1)
for p in pipes:
    with torch.cuda.stream(p[3]):
        if p[4]: jpeg.encode_end(p[0]); p[4] = False
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"step {dt / K * 1e3:.3f} ms; host time in begin {tb / K * 1e3:.3f} ms per step, in end (waiting) {te / K * 1e3:.3f} ms per step")
