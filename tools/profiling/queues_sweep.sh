#!/bin/bash
run() { env $1 python3 bench.py --no-cpu-baseline --no-verify --steps 12 --warmup 4 $2 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$1] [$2]:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'], d['hysteresis']['misses'])"; }
for rep in 0 1; do
run "GPU_MAX_HW_QUEUES=16" ""
run "GPU_MAX_HW_QUEUES=24" ""
run "GPU_MAX_HW_QUEUES=32" ""
run "GPU_MAX_HW_QUEUES=8" ""
done
