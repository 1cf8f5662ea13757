"""diagnostic (GPU box): the colour-plane stage of a blocking, profiled call on context 0 as a function of how many OTHER contexts
(with their own outputs and workspace) exist in the process, and of whether their memory is still allocated."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, adaptive_edge_aware_jpeg_amd as A

B, H, W = 64, 2160, 3840
dev = torch.device("cuda", 0)
xs = [bench.synth_batch(torch, B, H, W, s, dev) for s in (20250718, 21250718)]
jpeg = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)), device=0)


def make(stream):
    with torch.cuda.stream(stream):
        ctx = jpeg._bind()
        plan = ctx.plan(B, H, W)
        out = (ctx.empty((B * plan.coeff_stride,), torch.int32), ctx.empty((B * plan.leaf_stride, 4), torch.int32),
               ctx.empty((B * plan.state_stride,), torch.uint8), ctx.empty((B, 3, 4), torch.int64))
        jpeg.encode_into(ctx, xs[0], plan, *out)
    return ctx, plan, out, stream


def stage_times(p, n=6):
    ctx, plan, out, stream = p
    acc = {}
    with torch.cuda.stream(stream):
        ctx.set_sub_batches(1)
        ctx.set_profiling(True)
        for i in range(n + 2):
            jpeg.encode_into(ctx, xs[i & 1], plan, *out)
            if i >= 2:
                for k, v in ctx.stage_ms().items():
                    acc[k] = acc.get(k, 0.0) + v / n
        ctx.set_profiling(False)
    return {k: round(acc[k], 3) for k in ("color_planes", "clahe_blur", "sobel_nms", "dct64")}


p0 = make(torch.cuda.current_stream(dev))
print("1 context                         :", stage_times(p0), flush=True)
others = [make(torch.cuda.Stream(device=dev))]
print("2 contexts                        :", stage_times(p0), flush=True)
print("   (stages timed on context 1)    :", stage_times(others[0]), flush=True)
others.append(make(torch.cuda.Stream(device=dev)))
print("3 contexts                        :", stage_times(p0), flush=True)
for o in others:
    o[0]._ws = None
del others
torch.cuda.empty_cache()
print("other contexts' memory released   :", stage_times(p0), flush=True)
print("memory allocated now (GB)         :", round(torch.cuda.memory_allocated() / 1e9, 1), "reserved", round(torch.cuda.memory_reserved() / 1e9, 1))
