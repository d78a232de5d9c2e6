#!/bin/bash
# A/B of network-kernel variants on ONE box:  tools/gpu_mlp_ab.sh name1 name2 ...  (libodk_var_<name>.so built by `make mlpvar NAME= VFLAGS=`; "base" = libodk.so)
# Two interleaved rounds of tools/gpu_mlp_bench.py per variant -> gpurun_out/mlp_ab.txt
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/mlp_ab.txt; : > $OUT
for rnd in 1 2; do
  for v in "$@"; do
    lib=$ROOT/open_duck_playground_amd/csrc/libodk_var_$v.so; [ $v = base ] && lib=$ROOT/open_duck_playground_amd/csrc/libodk.so
    echo -n "$v: " >> $OUT
    ODK_LIB=$lib python3 $ROOT/tools/gpu_mlp_bench.py 2>&1 | tail -1 >> $OUT
  done
done
cat $OUT
