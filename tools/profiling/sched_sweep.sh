#!/bin/bash
# schedule sweep of the pipelined bench: contexts (--pipeline) x sub-batches per call, optional library options
#   bash tools/profiling/sched_sweep.sh "<pipelines>" "<sub-batch counts>" [extra bench args]
pipes=${1:-"2 3 4"}; subs=${2:-"2 3 4 6 8"}; shift; shift
for p in $pipes; do for s in $subs; do
python3 bench.py --no-cpu-baseline --no-verify --steps 40 --warmup 6 --pipeline $p --sub-batches $s "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('pipeline $p sub-batches $s $*:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'])"
done; done
