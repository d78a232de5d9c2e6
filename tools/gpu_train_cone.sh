#!/bin/bash
# The reference's default command line with elliptic friction cones (runner --cone elliptic): tools/gpu_train_cone.sh TAG
#   -> gpurun_out/train_cone_TAG/elliptic/metrics.jsonl + wall time
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
OUT=$ROOT/gpurun_out/train_cone_$TAG
mkdir -p $OUT
t0=$(date +%s)
python -m open_duck_playground_amd.runner --output_dir $OUT/elliptic --task flat_terrain --num_timesteps 150000000 --cone elliptic > $OUT/elliptic.log 2>&1
rc=$?
t1=$(date +%s)
echo "elliptic rc $rc wall_s $((t1 - t0)) args --task flat_terrain --num_timesteps 150000000 --cone elliptic" >> $OUT/wall.txt
rm -f $OUT/elliptic/*.pt $OUT/elliptic/*.onnx $OUT/elliptic/events.out.*
cat $OUT/wall.txt
