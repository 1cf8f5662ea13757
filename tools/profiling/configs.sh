#!/bin/bash
# BASELINE.json configs 2..5 on one GPU (per-GPU share of the multi-GPU ones); prints value / ms / stages per config
run() { python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-verify "$@" 2>/dev/null | python3 -c "import json,sys;d=json.load(sys.stdin);print('$*', '->', d['value'],'MP/s', d['ms_per_step'],'ms (blocking calls', d['pipeline']['serial_ms_per_step'], 'ms)', {k:v['ms'] for k,v in d['stages'].items()}, d['hysteresis'], d['leaves_per_image'])"; }
run --batch 1 --height 1080 --width 1920 --steps 200 --warmup 20
run --batch 64 --height 1080 --width 1920
run --batch 64
run --batch 8 --height 4320 --width 7680 --space OKLAB --blocks 4 128
run --batch 16 --space ICtCp
run --batch 16 --space YCoCg
run --batch 64 --ingest u8
