#!/bin/bash
# diagnostic: rebuild canny.hip with different occupancy targets for the blur kernel on the GPU box and time it
cd adaptive_edge_aware_jpeg_amd/csrc
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -fvisibility=hidden"
for v in ${@:-4 5 6}; do
  /opt/rocm/bin/hipcc $FLAGS -DAEJ_BLUR_OCC=$v -c canny.hip -o canny.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libaejpeg_hip.so api.o color.o canny.o quadtree.o dct.o decode.o metrics.o
  (cd ../..; python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys;d=json.load(sys.stdin);print('OCC $v', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})")
done
