#!/usr/bin/env python3
"""One blocking call of a latency-sized image under rocprofv3 --kernel-trace: the launches in order, their durations and the gaps between
them (GPU box).   python3 tools/profiling/latency_trace.py [H W]   -> prints the chain of the LAST call
(run as: rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lat -- python3 tools/profiling/latency_trace.py run 1080 1920; then
 python3 tools/profiling/latency_trace.py show gpurun_out/lat)"""
import csv, glob, os, sys
if sys.argv[1] == "run":
    sys.path.insert(0, os.getcwd())
    import torch, bench
    import adaptive_edge_aware_jpeg_amd as A
    H, W = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda", 0)
    x = bench.synth_batch(torch, 1, H, W, 20250718, dev)
    codec = A.Jpeg(A.JpegCompressionSettings("YCbCr", (40, 80), (4, 64)))
    for _ in range(20):
        codec.compress_batch(x)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(200):
        codec.compress_batch(x)
    torch.cuda.synchronize()
    print("blocking call: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
else:
    f = glob.glob(sys.argv[2] + "/*/*kernel_trace.csv")[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aej::", "")) for r in csv.DictReader(open(f)) if "aej::" in r["Kernel_Name"]]
    rows.sort()
    # the last call = the launches after the last long gap... take the last N launches where N = launches per call
    names = [r[2] for r in rows]
    first = names[-1]
    # find period: distance between the last two occurrences of the colour kernel
    idx = [i for i, n in enumerate(names) if "k_color_planes" in n]
    per = idx[-1] - idx[-2]
    call = rows[idx[-2]:idx[-1]]
    t0 = call[0][0]
    tot = 0
    for i, (s, e, n) in enumerate(call):
        gap = s - call[i - 1][1] if i else 0
        tot += e - s
        print(f"{n[:44]:44s} start {(s - t0) / 1e3:7.1f} us  dur {(e - s) / 1e3:6.1f}  gap {gap / 1e3:5.1f}")
    print(f"{per} launches; kernels {tot / 1e3:.1f} us; first start to last end {(call[-1][1] - t0) / 1e3:.1f} us; call to call {(rows[idx[-1]][0] - t0) / 1e3:.1f} us")
