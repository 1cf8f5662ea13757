#!/bin/bash
# in-tree library against several variant libraries (build/variants/<name>), interleaved: bash tools/profiling/ab_many.sh <reps> <name> [<name> ...]
run() { env $1 python3 bench.py --no-cpu-baseline --no-verify --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$2]:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'], '; stages', {k: v['ms'] for k, v in d['stages'].items() if k in ('clahe_blur', 'sobel_nms', 'hysteresis', 'dct4', 'dct32', 'dct64')})"; }
reps=$1; shift
for rep in $(seq $reps); do
run "A=1" in-tree
for v in "$@"; do run "AEJ_LIBRARY=build/variants/$v/libaejpeg_hip.so" $v; done
done
