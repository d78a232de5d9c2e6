#!/bin/bash
# Kernel trace of the learner inside the full-PPO loop:  tools/gpu_learner_trace.sh TAG [env VAR=..]  -> gpurun_out/ltrace_TAG/{timeline.txt, stats.txt}
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/ltrace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/tools/gpu_train_bench.py flat_terrain_backlash 2 > $OUT/bench.json 2> $OUT/err.txt
python3 $ROOT/tools/learner_timeline.py $OUT/kt 300 > $OUT/timeline.txt 2>&1
python3 $ROOT/tools/kernel_stats_top.py $OUT/kt 25 > $OUT/stats.txt 2>&1
rm -rf $OUT/kt
