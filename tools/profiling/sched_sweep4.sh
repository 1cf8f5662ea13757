#!/bin/bash
run() { env $1 python3 bench.py --no-cpu-baseline --no-verify --steps 12 --warmup 4 $2 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$1] [$2]:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'], d['hysteresis'])"; }
for rep in 0 1; do
run "A=1" ""
run "AEJ_HYST_MARGIN=2" ""
run "AEJ_HYST_MARGIN=1" ""
run "A=1" "--sub-batches 5"
done
