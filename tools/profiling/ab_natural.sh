#!/bin/bash
# in-tree library against one variant on the natural-image bench: bash tools/profiling/ab_natural.sh <variant> [reps]
run() { env $1 python3 bench.py --data natural --no-cpu-baseline --no-verify --steps 20 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$2 natural]:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'], '; stages', {k: v['ms'] for k, v in d['stages'].items() if k.startswith('dct')})"; }
for rep in $(seq ${2:-2}); do run "A=1" in-tree; run "AEJ_LIBRARY=build/variants/$1/libaejpeg_hip.so" $1; done
