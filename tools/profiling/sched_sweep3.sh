#!/bin/bash
run() { env $1 python3 bench.py --no-cpu-baseline --no-verify --steps 12 --warmup 4 $2 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$1] [$2]:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'], d['hysteresis']['misses'])"; }
for rep in 0 1; do
for m in 0 1 2 4 3 5 7; do run "AEJ_STAGE_CHAIN=$m" ""; done
run "AEJ_STAGE_CHAIN=1 AEJ_SUB_CHAIN=1" ""
run "AEJ_STAGE_CHAIN=7 AEJ_SUB_CHAIN=1" ""
run "AEJ_STAGE_CHAIN=1" "--pipeline 2"
run "AEJ_STAGE_CHAIN=7" "--pipeline 4"
done
