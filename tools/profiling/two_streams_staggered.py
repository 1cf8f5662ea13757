"""diagnostic (GPU box): two half-batches on two streams / host threads, FREE-RUNNING and out of phase, against one full batch.
The in-phase experiment (two_streams.py) overlaps equal stages (HBM-bound with HBM-bound); here thread 2 starts `stagger` ms late
and neither thread waits for the other, so the colour stage of one half can run beside the blur stage of the other.
    python3 tools/profiling/two_streams_staggered.py [batch]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, adaptive_edge_aware_jpeg_amd as A
from adaptive_edge_aware_jpeg_amd._lib import Context

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = 2160, 3840
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
        parts.append((ctx, x[i * n:(i + 1) * n], plan, bufs, s))
    return parts


def free_run(parts, K, stagger_ms):
    def work(i, p):
        ctx, xs, plan, bufs, stream = p
        if i:
            time.sleep(i * stagger_ms * 1e-3)
        with torch.cuda.stream(stream):          # torch's current stream is per host thread
            for _ in range(K):
                jpeg.encode_into(ctx, xs, plan, *bufs)
    ts = [threading.Thread(target=work, args=(i, p)) for i, p in enumerate(parts)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


K = 24
parts = make(1)
free_run(parts, 3, 0)
dt = free_run(parts, K, 0) / K
print(f"one stream, full batch        : {dt * 1e3:.3f} ms per {B} images  {B * H * W / dt / 1e6:.0f} MP/s", flush=True)
del parts
for nsplit in (2, 3, 4):
    if B % nsplit:
        continue
    parts = make(nsplit)
    free_run(parts, 3, 0)
    for stagger in (0.0, 0.5, 1.0, 1.5, 2.5):
        total = free_run(parts, K, stagger)
        dt = total / K                      # every part encodes B / nsplit images K times: K full batches in `total`
        print(f"{nsplit} streams, stagger {stagger:3.1f} ms: {dt * 1e3:.3f} ms per {B} images  {B * H * W / dt / 1e6:.0f} MP/s", flush=True)
    del parts

# one CALL's worth: every part encodes once, all joined before the next round (what a library-internal split could reach)
print("single-round mode (join after every round):", flush=True)
import statistics
for nsplit in (2, 4, 8):
    if B % nsplit:
        continue
    parts = make(nsplit)
    free_run(parts, 2, 0)
    for stagger in (0.0, 0.2, 0.4, 0.7, 1.0):
        ts = [free_run(parts, 1, stagger) for _ in range(12)]
        print(f"{nsplit} streams, stagger {stagger:3.1f} ms: median {statistics.median(ts) * 1e3:.3f} ms, best {min(ts) * 1e3:.3f} ms per {B} images", flush=True)
    del parts
