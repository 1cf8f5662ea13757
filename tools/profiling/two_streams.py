"""diagnostic (GPU box): does encoding two half-batches on two streams / host threads beat one full batch?"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, adaptive_edge_aware_jpeg_amd as A
from adaptive_edge_aware_jpeg_amd._lib import Context

B, H, W = 64, 2160, 3840
dev = torch.device("cuda", 0)
x = bench.synth_batch(torch, B, H, W, 20250718, dev)
settings = A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64))
jpeg = A.Jpeg(settings, device=0)


def make(nsplit):
    parts = []
    for i in range(nsplit):
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            ctx = Context(0)
        bmin, bmax = jpeg._block_sizes[0], jpeg._block_sizes[-1]
        ctx.set_settings(settings.color_space, bmin, bmax, jpeg._qmats_blob())
        n = B // nsplit
        plan = ctx.plan(n, H, W)
        bufs = (ctx.empty((n * plan.coeff_stride,), torch.int32), ctx.empty((n * plan.leaf_stride, 4), torch.int32),
                ctx.empty((n * plan.state_stride,), torch.uint8), ctx.empty((n, 3, 4), torch.int64))
        parts.append((ctx, x[i * n:(i + 1) * n], plan, bufs))
    return parts


def run(parts):
    def work(p):
        ctx, xs, plan, bufs = p
        jpeg.encode_into(ctx, xs, plan, *bufs)
    if len(parts) == 1:
        work(parts[0])
        return
    ts = [threading.Thread(target=work, args=(p,)) for p in parts]
    for t in ts: t.start()
    for t in ts: t.join()


for nsplit in (1, 2, 4):
    parts = make(nsplit)
    for _ in range(3): run(parts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 8
    for _ in range(K): run(parts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"split {nsplit}: {dt * 1e3:.3f} ms/step  {B * H * W / dt / 1e6:.0f} MP/s", flush=True)
    del parts
