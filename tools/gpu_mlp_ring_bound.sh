#!/bin/bash
# What would a shared weight ring in LDS buy the forward launch AT BEST?  The diagnostic build's forward kernel with the weights read
# from LDS instead of global memory (no fill, no synchronisation: diag bit 5), and with each tile group's share of the ring's fill traffic added (bit 6),
# against the same build with the weights streamed from the L2 as shipped; two interleaved rounds on ONE box.   tools/gpu_mlp_ring_bound.sh
#   (make -C open_duck_playground_amd/csrc libodk_mlpdiag.so first; results of the diag variants are WRONG on purpose)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/mlp_ring_bound.txt; : > $OUT
for rnd in 1 2; do
  for d in 0 32 96 4; do
    echo -n "diag $d: " >> $OUT
    ODK_LIB=$ROOT/open_duck_playground_amd/csrc/libodk_mlpdiag.so ODK_MLP_DIAG=$d python3 $ROOT/tools/gpu_mlp_bench.py 2>&1 | tail -1 >> $OUT
  done
  echo -n "shipped: " >> $OUT
  python3 $ROOT/tools/gpu_mlp_bench.py 2>&1 | tail -1 >> $OUT
done
cat $OUT
