#!/bin/bash
# copy what full_cycle.sh left under gpurun_out/ into profiles/ (the judged copies): bash tools/profiling/collect.sh a|p|b [tag]
tag=${2:-r04}
case "$1" in
a) cp gpurun_out/prof_${tag}_serial_kernel_stats.csv profiles/${tag}_kernel_stats_serial.csv; cp gpurun_out/prof_${tag}_pipelined_kernel_stats.csv profiles/${tag}_kernel_stats_pipelined.csv
   cp gpurun_out/prof_${tag}_serial.json profiles/${tag}_bench_under_rocprof_serial.json; cp gpurun_out/prof_${tag}_pipelined.json profiles/${tag}_bench_under_rocprof_pipelined.json ;;
p) cp gpurun_out/hbm_traffic.json profiles/${tag}_hbm_traffic.json; cp gpurun_out/cycle_pmc_traffic.txt profiles/${tag}_hbm_traffic.txt
   for g in valu mfma; do cp gpurun_out/pmc_$g.json profiles/${tag}_pmc_$g.json; cp gpurun_out/cycle_pmc_$g.txt profiles/${tag}_pmc_$g.txt; done ;;
b) cp gpurun_out/cycle_bench.json profiles/${tag}_bench.json; cp gpurun_out/cycle_bench_natural.json profiles/${tag}_bench_natural.json; cp gpurun_out/cycle_configs.txt profiles/${tag}_configs.txt
   cp gpurun_out/cycle_bench_2rank_gloo.json profiles/${tag}_bench_2rank_gloo_one_gpu_rehearsal.json; cp gpurun_out/cycle_bench_1rank_rccl.json profiles/${tag}_bench_1rank_rccl.json ;;
esac
