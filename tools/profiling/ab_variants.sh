#!/bin/bash
# interleaved A / B of the in-tree library against variant builds with the serial stage times printed:
#   bash tools/profiling/ab_variants.sh <reps> "<variant> ..." [bench args]
reps=$1; variants=$2; shift; shift
run() { env $1 python3 bench.py --no-cpu-baseline --no-verify --steps 30 --warmup 6 "${@:2}" 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$1]', d['ms_per_step'], 'blocking', d['pipeline']['serial_ms_per_step'], {k: v['ms'] for k, v in d['stages'].items()})"; }
for rep in $(seq $reps); do
  run "A=1" "$@"
  for v in $variants; do run "AEJ_LIBRARY=build/variants/$v/libaejpeg_hip.so" "$@"; done
done
