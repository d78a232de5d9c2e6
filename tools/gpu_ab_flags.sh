#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for r in 1 2; do for l in libodk.so libodk_var_lsv.so libodk_var_slp.so libodk_var_both.so; do
  ODK_LIB=$ROOT/open_duck_playground_amd/csrc/$l python3 $ROOT/bench.py --task rough_terrain_backlash --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l rough', d['value'], d['ms_per_step'])"
done; done
