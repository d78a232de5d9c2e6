#!/bin/bash
# Throughput of the flat-floor step kernel against workgroups per CU (dev build libodk_occ.so: ODK_LDS_PAD adds dynamic LDS per workgroup).
# 20 396 B per workgroup -> 8 per CU; the pads below give 7, 6, 5, 4, 3, 2.   tools/gpu_occupancy_scan.sh > gpurun_out/occupancy_scan.txt
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
for pad in 0 2600 6400 11800 20000 33000 61000; do
  v=$(ODK_LIB=open_duck_playground_amd/csrc/libodk_occ.so ODK_LDS_PAD=$pad python bench.py --steps 300 --warmup 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "pad $pad bytes -> $(( 163840 / (20396 + pad) )) workgroups per CU: env-steps/s, ms/step = $v"
done
