#!/bin/bash
# contexts x sub-batches for the 8-bit ingest workload at the driver's --steps 20 --warmup 5 (GPU_MAX_HW_QUEUES=24): bash tools/profiling/ab_sched_u8.sh
for r in 1 2 3; do
for cfg in "--pipeline 3 --pipelined-sub-batches 0" "--pipeline 4" "--pipeline 6" "--pipeline 6 --pipelined-sub-batches 1" "--pipeline 8 --pipelined-sub-batches 1"; do
python3 bench.py --ingest u8 --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-other-configs $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('u8 $cfg:', d['ms_per_step'], 'blocking', d['pipeline']['serial_ms_per_step'], d['sub_batches']['pipelined_steps'])"
done; done
