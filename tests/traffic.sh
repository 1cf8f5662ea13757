#!/bin/bash
# HBM traffic per kernel from PMC counters (GPU box).  Two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do
# not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") of the bench workload; writes gpurun_out/hbm_traffic.json.
#   bash tests/traffic.sh [batch]
export TMPDIR=/tmp
B=${1:-64}
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch $B > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
done
python3 - $B <<'PY'
import csv, glob, json, sys, collections
B = int(sys.argv[1]); calls = 2      # warm-up + 1 timed step
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "aej::" in r["Kernel_Name"] and r["Counter_Name"] == c:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aej::", "")
            acc[name] += float(r["Counter_Value"])
    for k, v in acc.items():
        out[k][c + "_KB_per_encode"] = v / calls
for k, d in out.items():
    # gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM); calibrated here on
    # k_color_planes whose read volume is known exactly (B*H*W*12 bytes)
    d["read_bytes"] = d.get("FETCH_SIZE_KB_per_encode", 0) * 1024 * 2
    d["write_bytes"] = d.get("WRITE_SIZE_KB_per_encode", 0) * 1024
    d["hbm_bytes"] = d["read_bytes"] + d["write_bytes"]
json.dump({"batch": B, "height": 2160, "width": 3840, "note": "per aej_encode_batch call; reads = 2 x FETCH_SIZE KB (gfx950 correction), writes = WRITE_SIZE KB",
           "kernels": out}, open("gpurun_out/hbm_traffic.json", "w"), indent=1, sort_keys=True)
cal = next((d for k, d in out.items() if k.startswith("k_color_planes<0, 2, 2")), {})
print("calibration: k_color_planes read_bytes", cal.get("read_bytes"), "expected", B * 2160 * 3840 * 12)
for k, d in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes"]):
    print(f"{k:34s} read {d['read_bytes']/1e9:7.3f} GB  write {d['write_bytes']/1e9:7.3f} GB")
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
