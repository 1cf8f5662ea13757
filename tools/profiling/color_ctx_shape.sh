#!/bin/bash
# VERDICT r2 item 2 (GPU box): colour kernel in a 1-context and a 2-context process: library stage time, rocprofv3 duration, FETCH_SIZE
export TMPDIR=/tmp
R=$PWD
for n in 1 2; do
  python3 tools/profiling/color_ctx_shape.py $n
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ccs_kt$n -- python3 tools/profiling/color_ctx_shape.py $n > /dev/null 2>&1
  f=$(ls gpurun_out/ccs_kt$n/*/*kernel_stats.csv | head -1)
  echo "  rocprofv3 kernel-trace, $n context(s):"; python3 tools/profiling/kstats.py $f | grep -i "color\|clahe_blur" | head -3
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ccs_pmc$n -- python3 tools/profiling/color_ctx_shape.py $n > /dev/null 2>&1
  f=$(ls gpurun_out/ccs_pmc$n/*/*counter_collection.csv | head -1)
  python3 - $f $n <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "color_planes" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Dispatch_Id"]].append(float(r["Counter_Value"]))
v = [sum(x) for x in acc.values()]
print(f"  FETCH_SIZE of k_color_planes_strip, {sys.argv[2]} context(s): {len(v)} launches, mean {sum(v)/len(v)*1024*2/1e9:.3f} GB read per launch (2 x FETCH_SIZE KB)")
PY
  rm -rf gpurun_out/ccs_kt$n gpurun_out/ccs_pmc$n
done
