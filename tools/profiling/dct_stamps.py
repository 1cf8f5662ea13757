#!/usr/bin/env python3
"""Diagnostic (GPU box, variant library built with -DAEJ_X_STAMPS): average cycles per phase of k_dct_mfma's leaf loop.
    AEJ_LIBRARY=build/variants/stamps/libaejpeg_hip.so python3 tools/profiling/dct_stamps.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import adaptive_edge_aware_jpeg_amd as A
from adaptive_edge_aware_jpeg_amd._lib import load_library
x = bench.synth_batch(torch, 16, 2160, 3840, 20250718, torch.device("cuda", 0))
codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
for _ in range(3):
    codec.compress_batch(x)
torch.cuda.synchronize()
lib = load_library()
buf = np.zeros((2, 512, 12), np.int64)
lib.aej_debug_read_stamps.restype = ctypes.c_int
assert lib.aej_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
names = ["wait X + barrier", "descriptor chunk refill + acc init", "chain 1 (MFMA)", "P write + barrier", "chain 2 (MFMA)", "end wait (next X landed) + next descriptor", "leaves", "loop head",
         "quantise + zigzag scatter to LDS", "barrier", "copy-out (LDS -> global)", "issue prefetch DMA of the next X"]
for k, S in enumerate((32, 64)):
    d = buf[k]
    d = d[d[:, 6] > 0]
    n = d[:, 6].sum()
    print(f"k_dct_mfma<{S}>: {len(d)} workgroups sampled, {n / len(d):.1f} leaves each; cycles per leaf (s_memtime ticks):")
    tot = 0
    for i in (7, 0, 11, 1, 2, 3, 4, 8, 9, 10, 5):
        v = d[:, i].sum() / n
        tot += v
        print(f"   {names[i]:46s} {v:9.0f}")
    print(f"   {'total':46s} {tot:9.0f}   (two chains of {S // 2} MFMAs = {2 * (S // 2) * 64} pipe cycles)")
