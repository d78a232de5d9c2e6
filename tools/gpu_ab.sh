#!/bin/bash
# A/B of two builds of libodk.so on ONE box (boxes differ by ~1 %):  tools/gpu_ab.sh [tasks...]
# compares open_duck_playground_amd/csrc/libodk_old.so (copy of the previous build) with libodk.so, two interleaved rounds
# (ODK_AB_LIBS="libodk_old.so libodk_x.so libodk.so": any list of builds in csrc/).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TASKS=${@:-flat_terrain flat_terrain_backlash}
LIBS=${ODK_AB_LIBS:-libodk_old.so libodk.so}
for r in 1 2; do for l in $LIBS; do for t in $TASKS; do
  ODK_LIB=$ROOT/open_duck_playground_amd/csrc/$l python3 $ROOT/bench.py --task $t --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l $t', d['value'], d['ms_per_step'])"
done; done; done
