#!/bin/bash
# VALU / wait counters of the step kernel in two --pmc passes:  tools/gpu_pmc_quick.sh [task]
set -u
TASK=${1:-flat_terrain}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmcq_$TASK
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PB="python3 $ROOT/bench.py --task $TASK --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p1 -- $PB > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F32 --output-format csv -d $OUT/p2 -- $PB > /dev/null 2> $OUT/p2.err
python3 $ROOT/tools/pmc_summary.py $OUT/summary.json $OUT/p1 $OUT/p2
rm -rf $OUT/p1 $OUT/p2
