#!/usr/bin/env python3
"""Diagnostic (GPU box, variant library built with -DAEJ_X_BLUR_STAMPS): average cycles per phase of k_clahe_blur's tile loop, per wave.
    AEJ_LIBRARY=build/variants/stamps/libaejpeg_hip.so python3 tools/profiling/blur_stamps.py"""
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
buf = np.zeros((512, 4, 10), np.int64)
lib.aej_debug_read_blur_stamps.restype = ctypes.c_int
assert lib.aej_debug_read_blur_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
names = ["strip prologue (tables, row classes)", "tile prologue (column classes, packed LUTs, 2 barriers)", "stage A (CLAHE)", "prefetch issue + barrier", "stage B (Gaussian)",
         "barrier", "stage C (bilateral + histogram + store)", "tiles", "final barrier"]
for wave in range(4):
    d = buf[:, wave, :]
    d = d[d[:, 7] > 0]
    n = d[:, 7].sum()
    print(f"wave {wave}: {len(d)} workgroups sampled, {n / len(d):.1f} tiles each; cycles per 128 x 32 tile (s_memtime ticks = shader cycles / ?):")
    tot = 0
    for i in (0, 1, 2, 3, 4, 5, 6, 8):
        v = d[:, i].sum() / n
        tot += v
        print(f"   {names[i]:58s} {v:9.0f}")
    print(f"   {'total':58s} {tot:9.0f}")
