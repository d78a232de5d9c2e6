#!/bin/bash
# A/B of two builds of libodk.so on ONE box (boxes differ by ~1 %):  tools/gpu_ab.sh [tasks...]
# compares open_duck_playground_amd/csrc/libodk_old.so (copy of the previous build) with libodk.so, two interleaved rounds.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TASKS=${@:-flat_terrain flat_terrain_backlash}
for r in 1 2; do for l in libodk_old.so libodk.so; do for t in $TASKS; do
  ODK_LIB=$ROOT/open_duck_playground_amd/csrc/$l python3 $ROOT/bench.py --task $t --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l $t', d['value'], d['ms_per_step'])"
done; done; done
