#!/bin/bash
# the blocking-call figure beside the pipelined one for several context / sub-batch settings (what showed that late-created sub-batch streams run
# slower, profiles/r05_sched_sweep.txt): bash tools/profiling/ab_block.sh
for r in 1 2; do
for cfg in "--pipeline 3 --pipelined-sub-batches 0" "--pipeline 4" "--pipeline 4 --sub-batches 4" "--pipeline 4 --pipelined-sub-batches 4" "--pipeline 1"; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-other-configs $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('$cfg:', d['ms_per_step'], 'blocking', d['pipeline']['serial_ms_per_step'], d['sub_batches'])"
done; done
