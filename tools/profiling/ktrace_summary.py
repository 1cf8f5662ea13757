#!/usr/bin/env python3
"""Per-kernel durations of a pipelined bench run from a rocprofv3 kernel trace (GPU box):

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --timed-only --steps 6 --warmup 4
    python3 tools/profiling/ktrace_summary.py gpurun_out/kt <encode calls in the run> [serial_stats.csv]

Prints, per kernel: launches per call, average duration, summed duration per call, and -- with the kernel stats of a serial run --
the dilation factor under overlap; then the busy / idle split of the device timeline (union of kernel intervals).
"""
import collections
import csv
import glob
import sys


def short(n):
    return n.split("(")[0].replace("void ", "").replace("aej::", "").strip()[:40]


d, calls = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "aej::" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
per = collections.defaultdict(list)
iv = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    per[short(r["Kernel_Name"])].append(e - s)
    iv.append((s, e))
serial = {}
if len(sys.argv) > 3:
    for r in csv.DictReader(open(sys.argv[3])):
        if "aej::" in r["Name"]:
            serial[short(r["Name"])] = float(r["AverageNs"])
tot = 0.0
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    avg = sum(v) / len(v)
    tot += sum(v) / calls
    dil = f"  x{avg / serial[k]:.2f} of serial {serial[k] / 1e3:.0f} us" if k in serial else ""
    print(f"{k:40s} launches/call {len(v) / calls:7.1f}  avg {avg / 1e3:8.1f} us  sum/call {sum(v) / calls / 1e6:7.3f} ms{dil}")
iv.sort()
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = max(e for _, e in iv) - t0
print(f"summed kernel time per call {tot:.2f} ms; timeline span {span / 1e6:.2f} ms = {span / calls / 1e6:.3f} ms per call, busy {busy / span:.1%}")
