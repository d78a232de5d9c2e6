#!/bin/bash
# Instruction-fetch counters of the step kernel (two separate --pmc passes, nothing else traced):  tools/gpu_icache_pmc.sh [task]
set -u
TASK=${1:-flat_terrain}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/icache_$TASK
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PB="python3 $ROOT/bench.py --task $TASK --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/p1 -- $PB > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH --output-format csv -d $OUT/p2 -- $PB > /dev/null 2> $OUT/p2.err
rocprofv3 --pmc SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p3 -- $PB > /dev/null 2> $OUT/p3.err
python3 $ROOT/tools/pmc_summary.py $OUT/summary.json $OUT/p1 $OUT/p2 $OUT/p3
tail -3 $OUT/p1.err $OUT/p2.err $OUT/p3.err | grep -i "error\|invalid\|fail" | head
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
