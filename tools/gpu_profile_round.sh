#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/gpu_profile_round.sh TAG        -> gpurun_out/prof_TAG/...   (tools/collect_round_profiles.sh TAG copies it into profiles/TAG/)
# Per task (flat_terrain = the headline, flat_terrain_backlash, rough_terrain_backlash): a kernel trace, the TCC traffic counters, the SQ
# instruction / wait counters, the float32 operation mix and the active-lane counters -- every --pmc group its own run, never combined
# with a trace domain -- folded into pmc_summary_<task>.json and traffic_<task>.json (what bench.py reports as roofline.traffic).
set -u
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for TASK in flat_terrain flat_terrain_backlash rough_terrain_backlash; do
  SFX=""; [ $TASK != flat_terrain ] && SFX="_$TASK"
  B="python3 $ROOT/bench.py --task $TASK --no-cpu-baseline --no-secondary"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B --steps 100 --warmup 10 > $OUT/bench_under_rocprof$SFX.json 2> $OUT/kt$SFX.err
  cp $(ls $OUT/kt/*/*kernel_stats.csv | head -1) $OUT/kernel_stats$SFX.csv
  PB="$B --steps 20 --warmup 5"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- $PB > /dev/null 2> $OUT/p1$SFX.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p2 -- $PB > /dev/null 2> $OUT/p2$SFX.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p3 -- $PB > /dev/null 2> $OUT/p3$SFX.err
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F32 --output-format csv -d $OUT/p4 -- $PB > /dev/null 2> $OUT/p4$SFX.err
  # float32 operation mix (the FLOPs behind bench.py's valu_frac: 64 lanes x (2 FMA + ADD + MUL + TRANS) wave-instructions)
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/p5 -- $PB > /dev/null 2> $OUT/p5$SFX.err
  # active lanes: VALUUtilization = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) (rocprofiler-sdk counter_defs.yaml)
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p6 -- $PB > /dev/null 2> $OUT/p6$SFX.err
  python3 $ROOT/tools/pmc_summary.py $OUT/pmc_summary$SFX.json $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 > /dev/null
  python3 - $OUT/pmc_summary$SFX.json $OUT/traffic$SFX.json $TASK $TAG <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); task, tag = sys.argv[3], sys.argv[4]
g = lambda k: d.get(k, {}).get("mean")
f, w = (g("FETCH_SIZE") or 0) * 1024, (g("WRITE_SIZE") or 0) * 1024
flop = 64 * (2 * (g("SQ_INSTS_VALU_FMA_F32") or 0) + (g("SQ_INSTS_VALU_ADD_F32") or 0) + (g("SQ_INSTS_VALU_MUL_F32") or 0) + (g("SQ_INSTS_VALU_TRANS_F32") or 0))
lanes = (g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)) if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU") else None
json.dump({"hbm_bytes_per_launch": f + w, "fetch": f, "write": w, "valu_flop_per_launch": flop, "valu_active_lane_fraction": lanes,
           "note": f"rocprofv3 --pmc passes over `bench.py --task {task} --steps 20 --warmup 5` (8192 envs), tools/gpu_profile_round.sh {tag}: FETCH_SIZE / WRITE_SIZE in KB -> bytes, no "
                   "width correction (4-byte-per-lane accesses are uncalibrated in MI355X_MICROARCH.md; the guide's x2 on FETCH_SIZE is for 16-byte streams); valu_flop_per_launch = 64 lanes x "
                   "(2 FMA_F32 + ADD_F32 + MUL_F32 + TRANS_F32) wave-instructions: every lane of an issued instruction counted; valu_active_lane_fraction = "
                   "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): the share of those lanes that was switched on (cycle-weighted)"}, open(sys.argv[2], "w"), indent=1)
PY
  rm -rf $OUT/kt $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6
done
cd $ROOT
python3 bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default.err
for t in flat_terrain_backlash rough_terrain_backlash; do python3 bench.py --task $t --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $OUT/bench_$t.json; done
python3 bench.py --mode ppo --task flat_terrain_backlash 2>/dev/null | tail -1 > $OUT/bench_ppo_config3.json
python3 bench.py --mode ppo --task rough_terrain_backlash 2>/dev/null | tail -1 > $OUT/bench_ppo_config4.json
python3 bench.py --mode ppo --task flat_terrain 2>/dev/null | tail -1 > $OUT/bench_ppo_flat.json
ls -la $OUT
