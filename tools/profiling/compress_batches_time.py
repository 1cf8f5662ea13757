"""Jpeg.compress_batches(in_flight=4) on 64 x 4K float32 batches, ms per batch (developer measurement, GPU box; also the subject of an API trace:
rocprofv3 --hip-trace --stats -- python3 tools/profiling/compress_batches_time.py).  `loop` as the argument times compress_batch in a loop instead.
One shape per process on purpose: HIP deals streams onto hardware queues in creation order, and a process that has created the streams of several
shapes one after the other measures their collisions (6.1-6.3 ms for the same call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
import adaptive_edge_aware_jpeg_amd as A
dev = torch.device("cuda", 0)
xs = [bench.synth_batch(torch, 64, 2160, 3840, s, dev) for s in (1, 2)]
codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
mode = sys.argv[1] if len(sys.argv) > 1 else "gen"
n = 24
for rep in range(3):
    if mode == "loop":
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            e = codec.compress_batch(xs[i & 1])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"compress_batch in a loop: {(t1 - t0) / n * 1e3:.3f} ms per batch")
        continue
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for e in codec.compress_batches((xs[i & 1] for i in range(n)), in_flight=4, inputs_ready=(mode == "ready")):
        if mode == "keep":
            keep = e
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{mode}: compress_batches(in_flight=4): {(t1 - t0) / n * 1e3:.3f} ms per batch")
