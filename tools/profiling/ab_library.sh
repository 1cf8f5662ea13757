#!/bin/bash
# A / B of the in-tree library against variant builds (build/variants/<name>/libaejpeg_hip.so), interleaved:
#   bash tools/profiling/ab_library.sh "<variant> ..." [repetitions] [extra bench args, e.g. --data natural]
variants=$1; reps=${2:-5}; shift; shift      # "" = the in-tree library only
run() { env $1 python3 bench.py --no-cpu-baseline --no-verify --steps 40 --warmup 6 "${@:2}" 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$1]:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'], '; stages', {k: v['ms'] for k, v in d['stages'].items() if k in ('clahe_blur', 'sobel_nms', 'hysteresis', 'quadtree', 'dct4', 'dct32', 'dct64')})"; }
for rep in $(seq $reps); do
run "A=1" "$@"
for v in $variants; do run "AEJ_LIBRARY=build/variants/$v/libaejpeg_hip.so" "$@"; done
done
