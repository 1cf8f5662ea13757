"""manual (GPU box): the bench workload encoded repeatedly must give identical outputs (hysteresis chase passes race by design;
the fix-point must not depend on the interleaving).  python tools/profiling/determinism_gpu.py [reps]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                      # noqa: E402
import bench                                      # noqa: E402
import adaptive_edge_aware_jpeg_amd as A          # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, H, W = 32, 2160, 3840
x = bench.synth_batch(torch, B, H, W, 20250718, torch.device("cuda", 0))
codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
seen = set()
for r in range(reps):
    enc = codec.compress_batch(x)
    torch.cuda.synchronize()
    cnt = enc.counts.cpu().numpy()
    h = hashlib.sha256()
    h.update(cnt.tobytes())
    for b in range(B):
        for l in range(3):
            d = enc.layer(b, l)
            h.update(d["coeffs"].tobytes()); h.update(d["leaves"].tobytes()); h.update(d["states"].tobytes())
    seen.add(h.hexdigest())
    print("rep", r, h.hexdigest()[:16], flush=True)
print("deterministic" if len(seen) == 1 else f"NOT deterministic: {len(seen)} distinct results")
sys.exit(0 if len(seen) == 1 else 1)
