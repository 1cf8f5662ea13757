#!/bin/bash
# hysteresis variants (tools/profiling/variants.py build canny.hip ...): stage time serial, synthetic and natural
for v in in-tree "$@"; do
  L="A=1"; [ $v != in-tree ] && L="AEJ_LIBRARY=build/variants/$v/libaejpeg_hip.so"
  for data in synthetic natural; do
  env $L python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --data $data 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$v $data]', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'], 'verified', d['verified']['ok'], {k: v['ms'] for k, v in d['stages'].items() if k in ('hysteresis', 'quadtree', 'dct4')}, d['hysteresis']['tiles_through_the_work_queue_last_call'])"
  done
done
