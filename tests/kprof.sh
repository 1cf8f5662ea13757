#!/bin/bash
# per-kernel timing of one bench run under rocprofv3 (GPU box): bash tests/kprof.sh <tag> [bench args]
export TMPDIR=/tmp
tag=$1; shift
R=$PWD
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/prof_${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
python3 tests/kstats.py gpurun_out/prof_${tag}_kernel_stats.csv
python3 -c "import json;d=json.load(open('gpurun_out/prof_$tag.json'));print(d['value'],d['ms_per_step'],{k:v['ms'] for k,v in d['stages'].items()})"
