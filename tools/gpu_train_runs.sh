#!/bin/bash
# End-to-end training evidence (the reference's command lines through this build's runner): tools/gpu_train_runs.sh TAG
#   -> gpurun_out/train_TAG/{flat,backlash,rough,standing,rough_up_normals}/metrics.jsonl + wall times (checkpoints / ONNX files are deleted: only the metrics travel back)
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
OUT=$ROOT/gpurun_out/train_$TAG
mkdir -p $OUT
run() {  # name, args...
  local name=$1; shift
  local t0=$(date +%s)
  python -m open_duck_playground_amd.runner --output_dir $OUT/$name "$@" > $OUT/$name.log 2>&1
  local t1=$(date +%s)
  echo "$name wall_s $((t1 - t0)) args $*" >> $OUT/wall.txt
  rm -f $OUT/$name/*.pt $OUT/$name/*.onnx $OUT/$name/events.out.*
}
run flat --task flat_terrain --num_timesteps 150000000
run backlash --task flat_terrain_backlash --num_timesteps 40000000
run rough --task rough_terrain_backlash --num_timesteps 40000000
run standing --env standing --task flat_terrain --num_timesteps 40000000
run rough_up_normals --task rough_terrain_backlash --num_timesteps 150000000 --hfield_up_normals_only
cat $OUT/wall.txt
