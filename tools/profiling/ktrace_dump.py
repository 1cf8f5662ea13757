#!/usr/bin/env python3
"""Compact dump of a rocprofv3 kernel trace for offline study: ktrace_dump.py <trace dir> <out.csv>  -> start_us,end_us,queue,kind (all kernels)"""
import csv, glob, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from ktrace_timeline import short
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rd = list(csv.DictReader(open(f)))
qkey = "Queue_Id" if "Queue_Id" in rd[0] else "Stream_Id"
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey], short(r["Kernel_Name"]) if "aej::" in r["Kernel_Name"] else r["Kernel_Name"][:40].replace(",", ";")) for r in rd)
t0 = rows[0][0]
with open(sys.argv[2], "w") as o:
    for s, e, q, k in rows:
        o.write(f"{(s - t0) / 1e3:.1f},{(e - t0) / 1e3:.1f},{q},{k}\n")
