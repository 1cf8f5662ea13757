#!/bin/bash
# bench.py once per environment setting (GPU box): bash tools/profiling/envruns.sh "<bench args>" "VAR=val VAR2=val" "..." ; an empty string = defaults
args=$1; shift
for rep in 0 1; do
for e in "$@"; do
  env $e python3 bench.py --no-cpu-baseline --no-verify --steps 8 --warmup 2 $args 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$e] rep $rep:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'], {k:round(v['ms'],3) for k,v in d['stages'].items()})"
done
done
