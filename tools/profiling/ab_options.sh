#!/bin/bash
# interleaved A / B of aej_set_option settings with the serial stage times printed:  bash tools/profiling/ab_options.sh <reps> "<opts 1>" "<opts 2>" ... [-- bench args]
reps=$1; shift
sets=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done; [ "$1" == "--" ] && shift
for rep in $(seq $reps); do
  for set in "${sets[@]}"; do
    args=""; [ "$set" != "-" ] && for o in $set; do args="$args --option $o"; done
    python3 bench.py --no-cpu-baseline --no-verify --steps 40 --warmup 6 $args "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$set]', d['ms_per_step'], 'blocking', d['pipeline']['serial_ms_per_step'], {k: v['ms'] for k, v in d['stages'].items()})"
  done
done
