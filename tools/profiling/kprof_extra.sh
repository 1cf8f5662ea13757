#!/bin/bash
# per-kernel timing of tools/bench_extra.py under rocprofv3 (GPU box): bash tools/profiling/kprof_extra.sh <tag> [args]
export TMPDIR=/tmp
tag=$1; shift
R=$PWD
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profx_$tag -- python3 tools/bench_extra.py --steps 3 "$@" > gpurun_out/profx_$tag.json 2> gpurun_out/profx_$tag.err
f=$(ls gpurun_out/profx_$tag/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/profx_${tag}_kernel_stats.csv
rm -rf gpurun_out/profx_$tag
python3 tools/profiling/kstats.py gpurun_out/profx_${tag}_kernel_stats.csv
