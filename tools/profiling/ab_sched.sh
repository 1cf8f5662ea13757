#!/bin/bash
# contexts x sub-batches of the pipelined steps at the driver's --steps 20 --warmup 5, interleaved three times (export GPU_MAX_HW_QUEUES=24 first: six contexts
# x two sub-batches are 19 streams): bash tools/profiling/ab_sched.sh   -> profiles/r05_sched_sweep.txt
for r in 1 2 3; do
for cfg in "--pipeline 3 --pipelined-sub-batches 0" "--pipeline 4" "--pipeline 4 --pipelined-sub-batches 3" "--pipeline 6 --pipelined-sub-batches 2"; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-other-configs $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('$cfg:', d['ms_per_step'], 'blocking', d['pipeline']['serial_ms_per_step'], d['sub_batches']['pipelined_steps'])"
done; done
