#!/bin/bash
# PMC counters per kernel (GPU box): bash tests/kpmc.sh <tag> "<counters>"
export TMPDIR=/tmp
tag=$1; ctr=$2
R=$PWD
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch ${AEJ_PMC_BATCH:-16} > gpurun_out/pmc_$tag.json 2> gpurun_out/pmc_$tag.err
f=$(ls gpurun_out/pmc_$tag/*/*counter_collection.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "aej::" not in k: continue
    acc[k[:44]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k, {c: f"{v:.3g}" for c, v in d.items()})
PY

rm -rf gpurun_out/pmc_$tag
