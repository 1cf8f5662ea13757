#!/bin/bash
# a live RCCL communicator and the hardware queues: one GPU, one rank, interleaved
#   no group | group, bench.py's defaults (24 queues, communicator created after our streams) | communicator first, 16 / 24 queues | ours first, 16 queues
run() { env $1 python3 bench.py --no-cpu-baseline --no-verify --steps 30 --warmup 6 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$1]:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'], (d['ranks'].get('collectives') or {}).get('backend'))"; }
for rep in 1 2; do
run "A=1"
run "AEJ_BENCH_FORCE_DIST=1 MASTER_PORT=29521"
run "AEJ_BENCH_FORCE_DIST=1 MASTER_PORT=29522 AEJ_BENCH_NCCL_EAGER=1 GPU_MAX_HW_QUEUES=16"
run "AEJ_BENCH_FORCE_DIST=1 MASTER_PORT=29523 AEJ_BENCH_NCCL_EAGER=1 GPU_MAX_HW_QUEUES=24"
run "AEJ_BENCH_FORCE_DIST=1 MASTER_PORT=29524 GPU_MAX_HW_QUEUES=16"
run "AEJ_BENCH_FORCE_DIST=1 MASTER_PORT=29525 GPU_MAX_HW_QUEUES=32"
done
