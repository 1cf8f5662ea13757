run() { env $1 python3 bench.py --no-cpu-baseline --no-verify --steps 12 --warmup 4 $2 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('[$1] [$2]:', d['ms_per_step'], 'ms/step; blocking', d['pipeline']['serial_ms_per_step'])"; }
for rep in 0 1; do
run "A=1" ""
run "A=1" "--sub-batches 3"
run "A=1" "--sub-batches 5"
run "A=1" "--sub-batches 6"
run "A=1" "--sub-batches 8"
run "A=1" "--pipeline 4"
run "A=1" "--pipeline 2"
done
