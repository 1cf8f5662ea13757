#!/usr/bin/env python3
"""PMC summaries of the bench workload per kernel (GPU box).  Each counter group is its own rocprofv3 pass with --kernel-trace
only (never combined with other trace domains), of `python3 bench.py --timed-only --pipeline 1 --sub-batches 1` (blocking calls on
one context, no sub-batches: counters are per kernel, and every kernel is launched a known number of times); values are divided by
the number of aej_encode_batch calls.

    python3 tools/profiling/pmc.py traffic [bench args]   -> gpurun_out/hbm_traffic.json   (FETCH_SIZE pass + WRITE_SIZE pass)
    python3 tools/profiling/pmc.py valu    [bench args]   -> gpurun_out/pmc_valu.json      (SQ instruction / LDS / busy counters)
    python3 tools/profiling/pmc.py mfma    [bench args]   -> gpurun_out/pmc_mfma.json      (MFMA busy, VMEM / LDS wait counters)

Copy the JSON to be judged into profiles/ (bench.py reads profiles/r02_hbm_traffic.json and profiles/r02_pmc_valu.json and labels
them with the commit they were taken at).  HBM bytes follow MI355X_MICROARCH.md's gfx950 correction: reads = 2 x FETCH_SIZE (the
counter tallies 128-byte requests at 64 B), writes = WRITE_SIZE, both reported in KB by rocprofv3.  The guide calibrates that factor for
16-byte-per-lane streaming only; profiles/r05_counter_calibration.txt (tools/profiling/counter_calibration.py) extends it to every access
shape of this library: 16 / 8 / 4 / 2 / 1 B per lane and the Sobel kernel's 64-byte row segments all read FETCH_SIZE = bytes / 2, stores of
16 / 4 / 2 / 1 B per lane WRITE_SIZE = bytes (+ <= 1.3 %).
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
GROUPS = {
    "traffic": [["FETCH_SIZE"], ["WRITE_SIZE"]],
    "valu": [["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT"],
             ["SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "SQ_INSTS_SMEM"]],
    "mfma": [["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
             ["SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"],
             ["GRBM_GUI_ACTIVE"]],      # with the dispatch's duration: the clock the kernel really ran at (busy cycles are fractions of THAT, not of the 2.4 GHz peak)
}


def short(name):
    return name.split("(")[0].replace("void ", "").replace("aej::", "").strip()


def one_pass(counters, bench_args, tag):
    out_dir = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
    shutil.rmtree(out_dir, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out_dir, "--", "python3", os.path.join(ROOT, "bench.py"),
                                                "--timed-only", "--steps", "2", "--warmup", "1", "--pipeline", "1", "--sub-batches", "1"] + bench_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        sys.stderr.write(r.stderr[-3000:])
        raise SystemExit(f"rocprofv3 pass {counters} failed ({r.returncode})")
    info = json.loads(line[-1])
    f = glob.glob(os.path.join(out_dir, "*", "*counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    seen = set()
    for row in csv.DictReader(open(f)):
        if "aej::" not in row["Kernel_Name"]:
            continue
        k = short(row["Kernel_Name"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key)
            launches[k] += 1
            if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                acc[k]["duration_ns_with_" + counters[0]] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    shutil.rmtree(out_dir, ignore_errors=True)
    return info, acc, launches


def src_hash():
    sys.path.insert(0, ROOT)
    import bench
    return bench.source_hash()


def main():
    mode = sys.argv[1]
    bench_args = sys.argv[2:]
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("AEJ_HEAD") or None
    kernels = collections.defaultdict(dict)
    info = None
    for i, counters in enumerate(GROUPS[mode]):
        info, acc, launches = one_pass(counters, bench_args, f"{mode}{i}")
        calls = info["encode_calls"]
        for k, d in acc.items():
            kernels[k]["launches_per_encode"] = launches[k] / calls
            for c, v in d.items():
                kernels[k][c + "_per_encode"] = v / calls

    def arg(name, default, n=1):
        if name in bench_args:
            i = bench_args.index(name)
            vals = bench_args[i + 1:i + 1 + n]
            return vals[0] if n == 1 else vals
        return default

    meta = {"batch": int(arg("--batch", 64)), "height": int(arg("--height", 2160)), "width": int(arg("--width", 3840)), "space": arg("--space", "YCbCr"),
            "blocks": [int(v) for v in arg("--blocks", [4, 64], 2)], "head": head, "src_hash": src_hash(), "bench": info}
    if mode == "traffic":
        for k, d in kernels.items():
            d["read_bytes"] = d.get("FETCH_SIZE_per_encode", 0.0) * 1024 * 2
            d["write_bytes"] = d.get("WRITE_SIZE_per_encode", 0.0) * 1024
            d["hbm_bytes"] = d["read_bytes"] + d["write_bytes"]
        meta["note"] = "per aej_encode_batch call; reads = 2 x FETCH_SIZE KB (gfx950 correction), writes = WRITE_SIZE KB; separate --pmc passes"
        out = "hbm_traffic.json"
    elif mode == "valu":
        for k, d in kernels.items():
            d["valu_insts_per_launch"] = d.get("SQ_INSTS_VALU_per_encode", 0.0) / max(d["launches_per_encode"], 1e-9)
        meta["note"] = "wave-level instruction counts per aej_encode_batch call; SQ_ACTIVE_* / SQ_WAIT_* / SQ_*_CYCLES are quad-cycle units summed over waves"
        out = "pmc_valu.json"
    else:
        meta["note"] = "SQ_VALU_MFMA_BUSY_CYCLES counts cycles; the other SQ cycle counters quad-cycles (MI355X_MICROARCH.md)"
        out = "pmc_mfma.json"
    meta["kernels"] = kernels
    path = os.path.join(ROOT, "gpurun_out", out)
    json.dump(meta, open(path, "w"), indent=1, sort_keys=True)
    tot = sum(d.get("hbm_bytes", 0.0) for d in kernels.values())
    print(path, f"total HBM bytes per encode: {tot / 1e9:.3f} GB" if mode == "traffic" else "")
    for k, d in sorted(kernels.items()):
        print(f"{k[:48]:48s}", {c.replace('_per_encode', ''): f"{v:.4g}" for c, v in d.items()})


if __name__ == "__main__":
    main()
