#!/bin/bash
# per-kernel timing of one bench run under rocprofv3 (GPU box): bash tools/profiling/kprof.sh <tag> [bench args]
# -> gpurun_out/prof_<tag>_kernel_stats.csv (copy the ones to be judged into profiles/)
export TMPDIR=/tmp
tag=$1; shift
R=$PWD
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 bench.py --steps 4 --warmup 2 --timed-only "$@" > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/prof_${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
python3 tools/profiling/kstats.py gpurun_out/prof_${tag}_kernel_stats.csv
cat gpurun_out/prof_$tag.json
