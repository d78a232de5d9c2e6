#!/bin/bash
# Kernel trace of one rollout step inside the full-PPO loop: tools/gpu_rollout_trace.sh TAG -> gpurun_out/rtrace_TAG/timeline.txt
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/rtrace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $ROOT/tools/gpu_train_bench.py flat_terrain_backlash 2 > $OUT/bench.json 2> $OUT/err.txt
python3 $ROOT/tools/rollout_timeline.py $OUT/kt 45 > $OUT/timeline.txt 2>&1
rm -rf $OUT/kt
