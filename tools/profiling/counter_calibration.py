#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE against known byte counts, per access shape (GPU box): tools/ubench/counter_calibration.hip under
rocprofv3, one --pmc pass per counter (kernel trace only) plus one --kernel-trace --stats pass for the durations.

    hipcc -O3 --offload-arch=gfx950 tools/ubench/counter_calibration.hip -o tools/ubench/counter_calibration.out     (here or on the box)
    python3 tools/profiling/counter_calibration.py > gpurun_out/counter_calibration.txt      -> copy to profiles/rNN_counter_calibration.txt

Prints, per kernel: bytes really touched per launch, the counter in bytes (rocprofv3 reports KB), counter / bytes, and the factor
tools/profiling/pmc.py must multiply the counter by.  The factors are also written as JSON (gpurun_out/counter_factors.json).
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "tools", "ubench", "counter_calibration.out")


def norm(name):
    return name.split("(")[0].replace("void ", "").replace(" ", "_").strip()


def run_pass(extra, tag):
    out_dir = os.path.join(ROOT, "gpurun_out", f"cal_{tag}")
    shutil.rmtree(out_dir, ignore_errors=True)
    cmd = ["rocprofv3"] + extra + ["--kernel-trace", "--output-format", "csv", "-d", out_dir, "--", EXE, "3"]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-3000:])
        raise SystemExit(f"rocprofv3 {extra} failed ({r.returncode})")
    return out_dir, r.stdout


def main():
    if not os.path.exists(EXE):
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", os.path.join(ROOT, "tools", "ubench", "counter_calibration.hip"), "-o", EXE])
    cases = {}
    counters = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        d, out = run_pass(["--pmc", cname], cname)
        for ln in out.splitlines():
            if ln.startswith("CASE "):
                _, k, b, rw = ln.split()
                cases[k] = (float(b), rw)
        acc, n = collections.defaultdict(float), collections.Counter()
        seen = set()
        for row in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0])):
            k = norm(row["Kernel_Name"])
            if row["Counter_Name"] != cname:
                continue
            acc[k] += float(row["Counter_Value"]) * 1024.0
            key = (row.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key)
                n[k] += 1
        counters[cname] = {k: acc[k] / n[k] for k in acc}
        shutil.rmtree(d, ignore_errors=True)
    d, _ = run_pass(["--stats"], "stats")
    dur = {}
    for row in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))[0])):
        dur[norm(row["Name"])] = float(row["AverageNs"]) * 1e-9
    shutil.rmtree(d, ignore_errors=True)

    print("access shape (kernel of tools/ubench/counter_calibration.hip)      bytes touched    counter (B)   counter/bytes   factor    GB/s")
    factors = {}
    for k, (b, rw) in cases.items():
        cname = "FETCH_SIZE" if rw == "R" else "WRITE_SIZE"
        hit = [v for kk, v in counters[cname].items() if kk == k]
        t = [v for kk, v in dur.items() if kk == k]
        if not hit:
            print(f"{k:60s} no counter row")
            continue
        ratio = hit[0] / b
        factors[k] = {"counter": cname, "bytes": b, "counter_bytes": hit[0], "factor": 1.0 / ratio, "seconds": t[0] if t else None}
        print(f"{cname[:5]} {k:56s} {b:14.0f} {hit[0]:14.0f} {ratio:10.3f} {1.0 / ratio:10.3f} {b / t[0] / 1e9 if t else 0:9.0f}")
    json.dump(factors, open(os.path.join(ROOT, "gpurun_out", "counter_factors.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
