#!/bin/bash
# the profile set behind DESIGN.md section 8, on the GPU box (outputs under gpurun_out/; copy the judged ones into profiles/):
#   bash tools/profiling/full_cycle.sh a    kernel stats (blocking calls, pipelined)
#   bash tools/profiling/full_cycle.sh p    the three PMC groups (separate rocprofv3 --pmc passes, kernel trace only)
#   bash tools/profiling/full_cycle.sh b    bench lines (default, natural images), BASELINE configurations, two-rank rehearsal
tag=${2:-r04}
case "$1" in
a)
  bash tools/profiling/kprof.sh ${tag}_serial --pipeline 1 --sub-batches 1 > gpurun_out/cycle_kprof_serial.log 2>&1; echo "kprof serial rc=$?"
  bash tools/profiling/kprof.sh ${tag}_pipelined > gpurun_out/cycle_kprof_pipelined.log 2>&1; echo "kprof pipelined rc=$?"
  ;;
p)
  for g in traffic valu mfma; do python3 tools/profiling/pmc.py $g > gpurun_out/cycle_pmc_$g.txt 2>&1; echo "pmc $g rc=$?"; done
  ;;
b)
  python3 bench.py --steps 20 > gpurun_out/cycle_bench.json 2> gpurun_out/cycle_bench.err; echo "bench rc=$?"
  python3 bench.py --steps 20 --data natural --no-cpu-baseline > gpurun_out/cycle_bench_natural.json 2> gpurun_out/cycle_bench_natural.err; echo "bench natural rc=$?"
  bash tools/profiling/configs.sh > gpurun_out/cycle_configs.txt 2>&1; echo "configs rc=$?"
  AEJ_BENCH_BACKEND=gloo AEJ_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 2 --steps 6 --batch 32 --no-cpu-baseline > gpurun_out/cycle_bench_2rank_gloo.json 2> gpurun_out/cycle_bench_2rank_gloo.err; echo "two ranks rc=$?"
  # one rank, but with a process group: the barriers and counter collectives run through RCCL (backend "nccl") on this box's one GPU
  AEJ_BENCH_FORCE_DIST=1 MASTER_PORT=29517 python3 bench.py --steps 20 --no-cpu-baseline > gpurun_out/cycle_bench_1rank_rccl.json 2> gpurun_out/cycle_bench_1rank_rccl.err; echo "one rank over RCCL rc=$?"
  ;;
esac
