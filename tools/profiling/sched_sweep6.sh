#!/bin/bash
# scheduling sweep for a variant library: bash tools/profiling/sched_sweep6.sh <variant>
run() { env $1 python3 bench.py --no-cpu-baseline --no-verify --steps 16 --warmup 4 $2 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$1] [$2]:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'])"; }
L="AEJ_LIBRARY=build/variants/$1/libaejpeg_hip.so"
run "A=1" ""
run "$L" ""
for p in 2 3 4; do for s in 2 3 6; do run "$L" "--pipeline $p --sub-batches $s"; done; done
run "A=1" ""
run "$L" ""
