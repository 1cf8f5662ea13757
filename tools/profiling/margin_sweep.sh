#!/bin/bash
# what a spare (empty) hysteresis pass per sub-batch chain costs the pipelined step: AEJ_HYST_MARGIN sweep
run() { env $1 python3 bench.py --no-cpu-baseline --no-verify --steps 30 --warmup 8 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$1]:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'], d['hysteresis'])"; }
for rep in 0 1; do for m in 0 1 2 3 5 8; do run "AEJ_HYST_MARGIN=$m"; done; done
