#!/usr/bin/env python3
"""Timing of the rows built beside the encode hot path: decode, 8-bit ingest, evaluation metrics, and the DROP-IN call end
to end -- `Jpeg.compress(Image) -> bytes` / `Jpeg.compress_many(batch)`: GPU pass, device-to-host copy of the coefficients, the per-layer
zlib level-9 streams of the container on host threads (src/jpeg/jpeg.py:588-590), bytes out -- with the split and the host cores used.
Diagnostic companion of bench.py (same synthetic 4K images); prints one JSON object.  GPU only.

    python tools/bench_extra.py [--batch 32] [--steps 5] [--e2e-images 16]
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


def end_to_end(torch, bench, A, codec, args, dev, H, W):
    """`compress_many` (and `compress` of one image) timed piece by piece: what a caller of the reference's API gets per second, and where
    the time goes once the hot path is on the GPU (SURVEY.md 8f-1 predicted the host deflate)."""
    import os
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    n = args.e2e_images
    make = bench.synth_batch if args.data == "synthetic" else bench.natural_batch
    x = make(torch, n, H, W, 777, dev)
    mp = n * H * W / 1e6
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    res = {"images": n, "data": args.data, "host_cpus": os.cpu_count(), "usable_cpus": cores}
    codec.compress_batch(x)                                   # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc = codec.compress_batch(x)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    layers = [[enc.layer(b, l) for l in range(3)] for b in range(n)]          # device -> host, per layer (what compress_many does)
    t_d2h = time.perf_counter() - t0
    raw = sum(L["coeffs"].nbytes for im in layers for L in im)
    for workers in sorted({1, min(16, cores), min(64, cores)}):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=workers) as ex:
            comp = list(ex.map(lambda L: len(zlib.compress(L["coeffs"].tobytes(), 9)), [L for im in layers for L in im]))
        dt = time.perf_counter() - t0
        res[f"zlib9_{workers}_threads"] = {"s": round(dt, 3), "MP/s": round(mp / dt, 1), "input_MB/s": round(raw / dt / 1e6, 1)}
    res["coefficient_bytes"] = raw
    res["compressed_bytes"] = int(sum(comp))
    res["gpu_pass"] = {"s": round(t_gpu, 4), "MP/s": round(mp / t_gpu, 1)}
    res["d2h_per_layer_copies"] = {"s": round(t_d2h, 3), "GB/s": round(raw / t_d2h / 1e9, 2)}
    workers = min(64, cores)
    t0 = time.perf_counter()
    blobs_default = codec.compress_many(x, extension=".png")                  # the default worker count (round 5: the usable cores)
    res["compress_many_default_workers"] = {"s": round(time.perf_counter() - t0, 3), "MP/s": round(mp / (time.perf_counter() - t0), 1)}
    t0 = time.perf_counter()
    blobs = codec.compress_many(x, extension=".png", workers=workers)
    assert blobs == blobs_default
    dt = time.perf_counter() - t0
    res["compress_many"] = {"s": round(dt, 3), "MP/s": round(mp / dt, 1), "workers": workers, "bytes_out": int(sum(len(b) for b in blobs)),
                            "bits_per_pixel": round(8.0 * sum(len(b) for b in blobs) / (n * H * W), 3)}
    for level in (1, 6):                                      # opt-in: same container, lower deflate effort (the reference's decoder reads any level)
        t0 = time.perf_counter()
        alt = codec.compress_many(x, extension=".png", workers=workers, zlib_level=level)
        dta = time.perf_counter() - t0
        res[f"compress_many_zlib_level_{level}"] = {"s": round(dta, 3), "MP/s": round(mp / dta, 1), "bytes_out": int(sum(len(b) for b in alt)),
                                                    "bits_per_pixel": round(8.0 * sum(len(b) for b in alt) / (n * H * W), 3)}
    # opt-in GPU entropy stage: LZ77 (hash chains over the 32 KiB window) + per-layer dynamic Huffman on the device, only compressed bytes cross to the host
    import zlib as _z
    enc2 = codec.compress_batch(x)
    codec.deflate_batch(enc2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    streams = codec.deflate_batch(enc2)
    t_def = time.perf_counter() - t0
    assert _z.decompress(streams[0][0]) == enc2.layer(0, 0)["coeffs"].tobytes()
    runs = []
    for _ in range(5):                                        # one call is ~10 ms of mostly host work: the fastest of five, all five reported
        t0 = time.perf_counter()
        gpu_blobs = codec.compress_many(x, extension=".png", entropy="gpu")
        runs.append(time.perf_counter() - t0)
    dtg = min(runs)
    res["compress_many_gpu_entropy"] = {"s": round(dtg, 4), "MP/s": round(mp / dtg, 1), "s_of_five_calls": [round(r, 4) for r in runs],
                                        "bytes_out": int(sum(len(b) for b in gpu_blobs)),
                                        "bits_per_pixel": round(8.0 * sum(len(b) for b in gpu_blobs) / (n * H * W), 3),
                                        "deflate_batch_s": round(t_def, 4), "note": "aej_deflate_histogram + host table helper + aej_deflate_batch + device-side compaction + one D2H of the compressed bytes; every container below is read back by the decoder",
                                        "vs_zlib9_bytes": round(sum(len(b) for b in gpu_blobs) / max(1, sum(len(b) for b in blobs)), 4)}
    codec.decompress(gpu_blobs[0])                           # the GPU-written container through the decode path (zlib.decompress inside)
    img = A.Image.from_array(x[0].cpu().numpy(), (H, W, 3), ".png")
    codec.compress(img)
    t0 = time.perf_counter()
    one = codec.compress(img)
    dt1 = time.perf_counter() - t0
    assert one == blobs[0]
    res["compress_one_image"] = {"s": round(dt1, 3), "MP/s": round(H * W / 1e6 / dt1, 1), "note": "Jpeg.compress(Image): host->device copy, GPU pass, D2H, three zlib-9 streams on three threads (round 4: one thread, 3.54 s)"}
    z = res[f"zlib9_{workers}_threads"]["s"]
    res["share_of_compress_many"] = {"gpu_pass": round(t_gpu / dt, 3), "d2h": round(t_d2h / dt, 3), "zlib9": round(z / dt, 3)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--e2e-images", type=int, default=16, help="images of the end-to-end compress_many measurement (zlib-9 of ~50 MB per 4K image: seconds of host time)")
    ap.add_argument("--data", choices=["synthetic", "natural"], default="synthetic")
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
    out["drop_in_end_to_end"] = end_to_end(torch, bench, A, codec, args, dev, H, W)
    sc = A.EvaluationMetrics.batch(x, dec).cpu().numpy()
    out["scores_mean"] = {"psnr_dB": round(float(sc[:, 0].mean()), 3), "ssim": round(float(sc[:, 1].mean()), 5), "ms_ssim": round(float(sc[:, 2].mean()), 5)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
