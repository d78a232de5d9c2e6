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
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_summary.json $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 > /dev/null
cd $ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for t in flat_terrain_backlash rough_terrain_backlash; do python3 bench.py --task $t --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$t.json; done
python3 tools/gpu_train_bench.py flat_terrain_backlash 8 2>/dev/null | tail -1 > $OUT/full_ppo_config3.json
python3 tools/gpu_train_bench.py rough_terrain_backlash 8 2>/dev/null | tail -1 > $OUT/full_ppo_config4.json
python3 tools/gpu_train_bench.py flat_terrain 8 2>/dev/null | tail -1 > $OUT/full_ppo_flat.json
rm -rf $OUT/kt $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
ls -la $OUT
