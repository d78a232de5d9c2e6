#!/bin/bash
# Long runs of the reference's command lines with the round's final build: tools/gpu_train_long.sh TAG
#   the reference README's "current win" line (flat_terrain_backlash, 300 M steps) and a 1 G-step flat_terrain soak
#   -> gpurun_out/train_long_TAG/{win,soak}/metrics.jsonl + wall times
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
OUT=$ROOT/gpurun_out/train_long_$TAG
mkdir -p $OUT
run() {
  local name=$1; shift
  local t0=$(date +%s)
  python -m open_duck_playground_amd.runner --output_dir $OUT/$name "$@" > $OUT/$name.log 2>&1
  local rc=$?
  local t1=$(date +%s)
  echo "$name rc $rc wall_s $((t1 - t0)) args $*" >> $OUT/wall.txt
  rm -f $OUT/$name/*.pt $OUT/$name/*.onnx $OUT/$name/events.out.*
}
run win --task flat_terrain_backlash --num_timesteps 300000000
run soak --task flat_terrain --num_timesteps 1000000000
cat $OUT/wall.txt
