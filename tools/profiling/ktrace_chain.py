#!/usr/bin/env python3
"""Where do the chains wait?  (GPU box; same trace as ktrace_timeline.py)

    python3 tools/profiling/ktrace_chain.py gpurun_out/kt

Per hardware queue (= one sub-batch stream) the aej kernels in order; for the steady-state part prints, per stage, the average duration and the
average gap between the previous kernel of the same queue ending and this one starting, and for the colour stage also the gap since the
previous colour kernel (of any queue) ended -- the stagger chain.  Then a Gantt-like listing of a few consecutive sub-batches.
"""
import collections
import csv
import glob
import sys

from ktrace_timeline import short  # noqa: E402  (same directory)

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rd = list(csv.DictReader(open(f)))
qkey = "Queue_Id" if "Queue_Id" in rd[0] else "Stream_Id"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r[qkey]) for r in rd if "aej::" in r["Kernel_Name"]]
rows.sort()
cut = rows[int(len(rows) * 0.4)][0]
byq = collections.defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for q, ks in byq.items():
    for i, k in enumerate(ks):
        if k[0] < cut or i == 0:
            continue
        dur[k[2]].append(k[1] - k[0])
        gap[k[2]].append(k[0] - ks[i - 1][1])
print(f"{len(byq)} queues; steady-state averages per stage (us): duration, gap after the previous kernel of the same queue")
for kind in ("colour", "blur", "sobel", "hyst0", "hystN", "qt", "dct4", "dct8", "dct16", "dct32", "dct64", "other"):
    if dur[kind]:
        n = len(dur[kind])
        print(f"  {kind:7s} n {n:4d}  dur {sum(dur[kind]) / n / 1e3:8.1f}  gap {sum(gap[kind]) / n / 1e3:8.1f}")
cols = [r for r in rows if r[2] == "colour" and r[0] >= cut]
g = [cols[i][0] - cols[i - 1][1] for i in range(1, len(cols))]
p = [cols[i][0] - cols[i - 1][0] for i in range(1, len(cols))]
print(f"colour chain: {len(cols)} kernels, start-to-start {sum(p) / len(p) / 1e3:.1f} us, previous colour's end -> this start {sum(g) / len(g) / 1e3:.1f} us")
blurs = [r for r in rows if r[2] == "blur" and r[0] >= cut]
p = [blurs[i][0] - blurs[i - 1][0] for i in range(1, len(blurs))]
g = [blurs[i][0] - blurs[i - 1][1] for i in range(1, len(blurs))]
print(f"blur: start-to-start {sum(p) / len(p) / 1e3:.1f} us, previous blur's end -> this start {sum(g) / len(g) / 1e3:.1f} us (negative = overlap)")
# Gantt of the first 6 colour kernels after the cut and everything in their queues until the queue's next colour
t0 = cols[0][0]
print("listing (us from the first listed colour start): queue, stage, start, end")
for c in cols[:6]:
    ks = byq[c[3]]
    i = ks.index(c)
    line = []
    last = None
    for k in ks[i:]:
        if k[2] == "colour" and k is not c:
            break
        if k[2] in ("hystN", "other"):
            last = k
            continue
        line.append(f"{k[2]} {(k[0] - t0) / 1e3:.0f}-{(k[1] - t0) / 1e3:.0f}")
    print(f"  q{c[3]}: " + "  ".join(line))
