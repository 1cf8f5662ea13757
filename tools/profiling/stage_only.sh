#!/bin/bash
# serial stage times of variant libraries (diagnostic builds, e.g. a kernel with its stores or its loads taken out): bash tools/profiling/stage_only.sh <variant> ...
for v in in-tree "$@"; do
  L="A=1"; [ $v != in-tree ] && L="AEJ_LIBRARY=build/variants/$v/libaejpeg_hip.so"
  env $L python3 bench.py --no-cpu-baseline --no-verify --steps 6 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$v]', {k: v['ms'] for k, v in d['stages'].items() if k.startswith('dct')})"
done
