#!/usr/bin/env python3
"""Steady-state timeline analysis of a pipelined bench run (GPU box):

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --timed-only --steps 12 --warmup 6
    python3 tools/profiling/ktrace_timeline.py gpurun_out/kt

Takes the last 60 % of the aej kernels (steady state), and prints: how long N kernels were running at once, per kernel kind the average
duration and what else was running beside it (time-weighted), and the idle share of the window.
"""
import collections
import csv
import glob
import sys


def short(n):
    n = n.split("(")[0].replace("void ", "").replace("aej::", "").strip()
    for k, v in (("k_color_planes", "colour"), ("k_clahe_blur", "blur"), ("k_sobel", "sobel"), ("k_hyst_pass0", "hyst0"), ("k_hyst_bulk", "hystB"), ("k_hyst_drain", "hystQ"),
                 ("k_qt_", "qt"), ("k_dct_mfma<64", "dct64"), ("k_dct_mfma<32", "dct32"), ("k_dct16", "dct16"), ("k_dct8", "dct8"), ("k_dct4", "dct4")):
        if n.startswith(k):
            return v
    return "other"


f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f)) if "aej::" in r["Kernel_Name"]]
rows.sort()
rows = rows[int(len(rows) * 0.4):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, k in rows:
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
running = collections.Counter()
conc_time = collections.Counter()
beside = collections.defaultdict(collections.Counter)
alone = collections.Counter()
last = t0
for t, d, k in ev:
    dt = t - last
    if dt > 0:
        n = sum(running.values())
        conc_time[n] += dt
        kinds = [x for x in running if running[x] > 0]
        for a in kinds:
            if n == running[a] and len(kinds) == 1:
                alone[a] += dt
            for b in kinds:
                if b != a or running[a] > 1:
                    beside[a][b] += dt
    running[k] += d
    last = t
span = t1 - t0
print(f"window {span / 1e6:.2f} ms, {len(rows)} kernels")
print("kernels running at once (share of the window):", {n: f"{v / span:.1%}" for n, v in sorted(conc_time.items())})
dur = collections.defaultdict(list)
for s, e, k in rows:
    dur[k].append(e - s)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    tot = sum(v)
    top = ", ".join(f"{b} {t / tot:.0%}" for b, t in beside[k].most_common(5))
    print(f"{k:7s} n {len(v):4d} avg {sum(v) / len(v) / 1e3:8.1f} us  total {tot / 1e6:7.2f} ms ({tot / span:.0%} of window)  alone {alone[k] / tot:.0%}  beside: {top}")
