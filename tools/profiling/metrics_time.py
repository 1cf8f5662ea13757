"""EvaluationMetrics.batch on 32 x 4K image pairs (developer measurement, GPU box): ms per call for PSNR / + SSIM / + MS-SSIM.
python3 tools/profiling/metrics_time.py [batch]     (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import adaptive_edge_aware_jpeg_amd as A
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
x = bench.synth_batch(torch, B, 2160, 3840, 20250718, dev)
codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
dec = codec.decompress_batch(codec.compress_batch(x))
for name, which in (("psnr", 1), ("psnr+ssim", 3), ("psnr+ssim+ms_ssim", 7)):
    A.EvaluationMetrics.batch(x, dec, which); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        sc = A.EvaluationMetrics.batch(x, dec, which)
    torch.cuda.synchronize()
    print(f"{name:20s} {(time.perf_counter() - t0) / 5 * 1e3:8.3f} ms per call of {B} pairs   scores[0] = {[round(float(v), 6) for v in sc[0].cpu()]}")
