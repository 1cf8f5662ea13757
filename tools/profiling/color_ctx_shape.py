"""diagnostic (GPU box): blocking, unsplit 64 x 4K calls on ONE context (on its own non-null stream) in a process that holds
N contexts in all (each with its own stream, outputs and workspace, each used once) -- the "process shape" question of VERDICT r2
item 2: does the colour-plane kernel run slower when a second context exists?

    python3 tools/profiling/color_ctx_shape.py N            -> library stage times (HIP events)
    rocprofv3 --kernel-trace --stats ... -- python3 tools/profiling/color_ctx_shape.py N
    rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/profiling/color_ctx_shape.py N
"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import adaptive_edge_aware_jpeg_amd as A
import bench

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B, H, W = 64, 2160, 3840
dev = torch.device("cuda", 0)
xs = [bench.synth_batch(torch, B, H, W, s, dev) for s in (20250718, 21250718)]
jpeg = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)), device=0)


def make(stream):
    with torch.cuda.stream(stream):
        ctx = jpeg._bind()
        ctx.set_sub_batches(1)
        plan = ctx.plan(B, H, W)
        out = (ctx.empty((B * plan.coeff_stride,), torch.int32), ctx.empty((B * plan.leaf_stride, 4), torch.int32),
               ctx.empty((B * plan.state_stride,), torch.uint8), ctx.empty((B, 3, 4), torch.int64))
        jpeg.encode_into(ctx, xs[0], plan, *out)
    return ctx, plan, out, stream


pipes = [make(torch.cuda.Stream(device=dev)) for _ in range(n_ctx)]
torch.cuda.synchronize()
ctx, plan, out, stream = pipes[0]
acc, n = {}, 6
with torch.cuda.stream(stream):
    ctx.set_profiling(True)
    for i in range(n + 2):
        jpeg.encode_into(ctx, xs[i & 1], plan, *out)
        if i >= 2:
            for k, v in ctx.stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v / n
    ctx.set_profiling(False)
print(f"{n_ctx} context(s): addresses in {xs[0].data_ptr():#x} norm/ws {ctx._ws.data_ptr():#x};",
      {k: round(acc[k], 3) for k in ("color_planes", "clahe_blur", "sobel_nms", "quadtree", "dct64")}, flush=True)
