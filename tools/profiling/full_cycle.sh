#!/bin/bash
# the profile set behind DESIGN.md section 7, in one go on the GPU box (outputs under gpurun_out/; copy the judged ones into profiles/):
#   bash tools/profiling/full_cycle.sh a    kernel stats (blocking calls, pipelined) + the three PMC groups
#   bash tools/profiling/full_cycle.sh b    bench lines (default, natural images), BASELINE configurations, fuzz
set -e
case "$1" in
a)
  bash tools/profiling/kprof.sh r04_serial --pipeline 1 --sub-batches 1 > gpurun_out/cycle_kprof_serial.log 2>&1; echo "kprof serial done"
  bash tools/profiling/kprof.sh r04_pipelined > gpurun_out/cycle_kprof_pipelined.log 2>&1; echo "kprof pipelined done"
  for g in traffic valu mfma; do python3 tools/profiling/pmc.py $g > gpurun_out/cycle_pmc_$g.txt 2>&1; echo "pmc $g done"; done
  ;;
b)
  python3 bench.py --steps 20 > gpurun_out/cycle_bench.json 2> gpurun_out/cycle_bench.err; echo "bench done"
  python3 bench.py --steps 20 --data natural > gpurun_out/cycle_bench_natural.json 2> gpurun_out/cycle_bench_natural.err; echo "bench natural done"
  bash tools/profiling/configs.sh > gpurun_out/cycle_configs.txt 2>&1; echo "configs done"
  python3 tests/manual/fuzz_gpu.py > gpurun_out/cycle_fuzz.txt 2>&1; tail -2 gpurun_out/cycle_fuzz.txt
  ;;
esac
