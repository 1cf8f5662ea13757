#!/bin/bash
# interleaved sweep of aej_set_option settings in the pipelined bench:  bash tools/profiling/option_sweep.sh <reps> "<opts of run 1>" "<opts of run 2>" ...
# (an option set is a space-separated list of name=value; "-" = the defaults)
reps=$1; shift
for rep in $(seq $reps); do
  for set in "$@"; do
    args=""; [ "$set" != "-" ] && for o in $set; do args="$args --option $o"; done
    python3 bench.py --no-cpu-baseline --no-verify --steps 40 --warmup 6 $args 2>/dev/null | python3 -c "import json,sys;d=json.load(sys.stdin);print('[$set]', d['ms_per_step'], 'blocking', d['pipeline']['serial_ms_per_step'])"
  done
done
