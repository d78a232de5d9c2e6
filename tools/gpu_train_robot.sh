#!/bin/bash
# A robot that is not the duck through the runner (reference README.md:74-85): tools/gpu_train_robot.sh TAG [xml] [timesteps]
#   -> gpurun_out/train_TAG/<robot>/metrics.jsonl + wall time (checkpoints / ONNX files are deleted: only the metrics travel back)
set -u
TAG=${1:-x}
XML=${2:-tests/assets/biped12.xml}
STEPS=${3:-100000000}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
OUT=$ROOT/gpurun_out/train_$TAG
mkdir -p $OUT
name=$(basename $XML .xml)
t0=$(date +%s)
python -m open_duck_playground_amd.runner --xml $XML --num_timesteps $STEPS --output_dir $OUT/$name > $OUT/$name.log 2>&1
rc=$?
t1=$(date +%s)
echo "$name rc $rc wall_s $((t1 - t0)) args --xml $XML --num_timesteps $STEPS" >> $OUT/wall.txt
ls $OUT/$name/*.onnx 2>/dev/null | wc -l | xargs echo "$name onnx_files" >> $OUT/wall.txt
rm -f $OUT/$name/*.pt $OUT/$name/*.onnx $OUT/$name/events.out.*
tail -5 $OUT/$name.log
cat $OUT/wall.txt
