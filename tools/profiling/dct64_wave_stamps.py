#!/usr/bin/env python3
"""Diagnostic (GPU box, variant library built with -DAEJ_X_STAMPS): cycles per phase of k_dct64_wave's leaf loop, per wave.
    AEJ_LIBRARY=build/variants/stamps/libaejpeg_hip.so python3 tools/profiling/dct64_wave_stamps.py"""
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
d = buf[1]
d = d[d[:, 6] > 0]
n = d[:, 6].sum()
names = {0: "descriptor (+ in this build: the wait for X)", 1: "issue the loads of X", 2: "drain of the vector memory queue", 3: "chain 1 (128 MFMAs)", 4: "chain 2 (128 MFMAs)", 5: "epilogue"}
print(f"k_dct64_wave: {len(d)} waves sampled, {n / len(d):.1f} leaves each; cycles per leaf (s_memtime ticks):")
tot = 0
for i in range(6):
    v = d[:, i].sum() / n
    tot += v
    print(f"   {names[i]:40s} {v:9.0f}")
print(f"   {'total':40s} {tot:9.0f}   (256 MFMAs = 16384 pipe cycles)")
print(f"   table set-up before the first leaf: {d[:, 7].mean():.0f} cycles per wave ({d[:, 7].mean() / (tot * n / len(d)) * 100:.1f} % of the wave's leaf time in this run)")
