#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/gpu_profile_round.sh TAG        -> gpurun_out/prof_TAG/{kernel_stats.csv, bench_under_rocprof.json, pmc_summary.json, bench_default.json, ...}
# Kernel trace and each --pmc group are separate runs (counters are never combined with other trace domains).
set -u
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cp $(ls $OUT/kt/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
PB="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc1 -- $PB > /dev/null 2> $OUT/pmc1.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc2 -- $PB > /dev/null 2> $OUT/pmc2.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc3 -- $PB > /dev/null 2> $OUT/pmc3.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F32 --output-format csv -d $OUT/pmc4 -- $PB > /dev/null 2> $OUT/pmc4.err
# float32 operation mix of the step kernel (the FLOPs behind bench.py's valu_frac: 64 lanes x (2 FMA + ADD + MUL + TRANS) wave-instructions)
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/pmc5 -- $PB > /dev/null 2> $OUT/pmc5.err
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_summary.json $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 > /dev/null
cd $ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for t in flat_terrain_backlash rough_terrain_backlash; do python3 bench.py --task $t --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$t.json; done
python3 bench.py --mode ppo --task flat_terrain_backlash 2>/dev/null | tail -1 > $OUT/bench_ppo_config3.json
python3 bench.py --mode ppo --task rough_terrain_backlash 2>/dev/null | tail -1 > $OUT/bench_ppo_config4.json
python3 bench.py --mode ppo --task flat_terrain 2>/dev/null | tail -1 > $OUT/bench_ppo_flat.json
# the same traffic / instruction counters for the height-field kernel (config 4's env step)
PR="python3 $ROOT/bench.py --task rough_terrain_backlash --steps 20 --warmup 5 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktr -- python3 $ROOT/bench.py --task rough_terrain_backlash --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_rough_under_rocprof.json 2> $OUT/ktr.err
cp $(ls $OUT/ktr/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_rough.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pr1 -- $PR > /dev/null 2> $OUT/pr1.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pr2 -- $PR > /dev/null 2> $OUT/pr2.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pr3 -- $PR > /dev/null 2> $OUT/pr3.err
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_summary_rough.json $OUT/pr1 $OUT/pr2 $OUT/pr3 > /dev/null
cd $ROOT
rm -rf $OUT/kt $OUT/ktr $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 $OUT/pr1 $OUT/pr2 $OUT/pr3
ls -la $OUT
