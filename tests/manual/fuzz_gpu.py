"""Randomised parity sweep (GPU box, manual): random sizes / colour spaces / block ranges / image kinds and small batches, whole encode
and decode against the oracle, and the opt-in GPU entropy stage against zlib.decompress.  python tests/manual/fuzz_gpu.py [n_cases] [seed]"""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import adaptive_edge_aware_jpeg_amd as A          # noqa: E402
from oracle import oracle as O                     # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.build()
spaces = ["YCbCr", "YCoCg", "YCoCg-R", "OKLAB", "ICtCp", "ICaCb", "JzAzBz"]
bad = 0
t0 = time.time()
for case in range(n_cases):
    space = spaces[rng.integers(len(spaces))]
    H, W = int(rng.integers(5, int(os.environ.get("AEJ_FUZZ_MAXH", "700")))), int(rng.integers(5, int(os.environ.get("AEJ_FUZZ_MAXW", "900"))))
    if rng.random() < 0.3:
        H, W = (H // 4 + 1) * 4, (W // 4 + 1) * 4
    lo = int(2 ** rng.integers(1, 5))
    hi = int(lo * 2 ** rng.integers(0, 5))
    hi = min(hi, 256)
    q = sorted(int(v) for v in rng.integers(5, 96, size=2))
    kind = ["mixed", "noise", "flat", "mixed"][rng.integers(4)]
    img = O.synth_image(H, W, int(rng.integers(1 << 30)), kind).astype(np.float32) / np.float32(255.0)
    if rng.random() < 0.3:            # arbitrary floats, some 8-bit levels
        img = np.where(rng.random((H, W, 1)) < 0.5, img, rng.random((H, W, 3), dtype=np.float32)).astype(np.float32)
    tag = f"case {case}: {space} {H}x{W} blocks ({lo},{hi}) q {tuple(q)} {kind}"
    try:
        codec = A.Jpeg(A.JpegCompressionSettings(space, tuple(q), (lo, hi)))
        nb = int(rng.integers(1, 4))          # the image rides in a batch of 1..3 (the others: flipped copies), at a random position
        pos = int(rng.integers(nb))
        batch = np.stack([img if b == pos else np.ascontiguousarray(img[::-1, ::-1]) for b in range(nb)])
        enc = codec.compress_batch(batch)
        ref = O.encode_image(img, space, tuple(q), (lo, hi))
        ok = True
        streams = codec.deflate_batch(enc, adaptive=bool(case & 1))
        for l in range(3):
            got = enc.layer(pos, l)
            ok &= zlib.decompress(streams[pos][l]) == got["coeffs"].tobytes()
            ok &= got["root_size"] == ref[l]["root_size"] and np.array_equal(got["states"], ref[l]["states"])
            ok &= np.array_equal(got["leaves"], ref[l]["leaves"]) and np.array_equal(got["coeffs"], ref[l]["coeffs"])
        dec = codec.decompress_batch(enc).cpu().numpy()[pos]
        want = O.decode_image(O.write_ajpg(ref, H, W, space, tuple(q), (lo, hi), ".png"))
        ok &= np.array_equal(dec, want, equal_nan=True)
        if not ok:
            bad += 1
            print("MISMATCH", tag, flush=True)
    except Exception as e:                     # noqa: BLE001
        bad += 1
        print("ERROR", tag, type(e).__name__, str(e)[:200], flush=True)
print(f"{n_cases - bad} of {n_cases} cases identical ({time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
