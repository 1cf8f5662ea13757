run() { env $1 python3 bench.py --no-cpu-baseline --no-verify --steps 40 --warmup 6 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$1]:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'], '; stages', {k: v['ms'] for k, v in d['stages'].items() if k in ('clahe_blur', 'sobel_nms', 'dct32', 'dct64')})"; }
for rep in $(seq ${2:-5}); do
run "A=1"
run "AEJ_LIBRARY=build/variants/${1:-head}/libaejpeg_hip.so"
done
