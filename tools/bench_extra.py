#!/usr/bin/env python3
"""Timing of the rows built beside the encode hot path: decode, 8-bit ingest, evaluation metrics.
Diagnostic companion of bench.py (same synthetic 4K images); prints one JSON object.  GPU only.

    python tools/bench_extra.py [--batch 32] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(torch, fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import bench
    import adaptive_edge_aware_jpeg_amd as A
    B, H, W = args.batch, 2160, 3840
    dev = torch.device("cuda", 0)
    x = bench.synth_batch(torch, B, H, W, 20250718, dev)
    x8 = (x * 255.0).round().to(torch.uint8)
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    mp = B * H * W / 1e6
    out = {"workload": f"{B} x {W}x{H} 'mixed' images, YCbCr, blocks 4-64, quality 40-80", "steps": args.steps}
    enc = codec.compress_batch(x)
    t = timed(torch, lambda: codec.compress_batch(x), args.steps)
    out["encode_f32"] = {"ms": round(t, 3), "MP/s": round(mp / t * 1e3, 1), "note": "compress_batch incl. output allocation"}
    t = timed(torch, lambda: codec.compress_batch(x8), args.steps)
    out["encode_u8_ingest"] = {"ms": round(t, 3), "MP/s": round(mp / t * 1e3, 1)}
    dec = codec.decompress_batch(enc)
    t = timed(torch, lambda: codec.decompress_batch(enc), args.steps)
    out["decode"] = {"ms": round(t, 3), "MP/s": round(mp / t * 1e3, 1), "note": "aej_decode_batch: dequantise + IDCT + merge + up-sample + inverse colour"}
    for name, which in (("psnr", 1), ("psnr+ssim", 3), ("psnr+ssim+ms_ssim", 7)):
        t = timed(torch, lambda: A.EvaluationMetrics.batch(x, dec, which), args.steps)
        out["metrics_" + name] = {"ms": round(t, 3), "MP/s": round(mp / t * 1e3, 1)}
    sc = A.EvaluationMetrics.batch(x, dec).cpu().numpy()
    out["scores_mean"] = {"psnr_dB": round(float(sc[:, 0].mean()), 3), "ssim": round(float(sc[:, 1].mean()), 5), "ms_ssim": round(float(sc[:, 2].mean()), 5)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
