#!/bin/bash
# Copies the evidence gathered on the GPU box (gpurun_out/, scratch) into profiles/<round>/ (tracked):
#   tools/collect_round_profiles.sh r2      after tools/gpu_profile_round.sh, gpu_learner_trace.sh, gpu_rollout_trace.sh,
#   gpu_substep_scan.py, gpu_train_runs.sh, gpu_phase_profile.py and gpu_icache_pmc.sh wrote their outputs
set -eu   # a missing input must fail loudly: stale evidence in profiles/ is worse than none
R=${1:-r2}; P=profiles/$R; G=gpurun_out/prof_$R
cp $G/kernel_stats.csv $G/bench_under_rocprof.json $G/pmc_summary.json $P/
cp $G/bench_default.json $P/bench_default_run.json
cp $G/bench_flat_terrain_backlash.json $G/bench_rough_terrain_backlash.json $G/full_ppo_config3.json $G/full_ppo_config4.json $G/full_ppo_flat.json $P/
cp gpurun_out/ltrace_${R}end/timeline.txt $P/learner_step_timeline.txt; cp gpurun_out/ltrace_${R}end/stats.txt $P/learner_kernel_stats_top.txt
cp gpurun_out/rtrace_${R}end/timeline.txt $P/rollout_step_timeline.txt
cp gpurun_out/substep_scan.txt $P/substep_scan.txt
cp gpurun_out/icache_flat_terrain/summary.json $P/icache_pmc_summary.json
cp gpurun_out/train_${R}end/flat/metrics.jsonl $P/train_flat_terrain_150M_metrics.jsonl
cp gpurun_out/train_${R}end/backlash/metrics.jsonl $P/train_config3_backlash_40M_metrics.jsonl
cp gpurun_out/train_${R}end/rough/metrics.jsonl $P/train_config4_rough_40M_metrics.jsonl
cp gpurun_out/train_${R}end/standing/metrics.jsonl $P/train_standing_40M_metrics.jsonl
cp gpurun_out/phase_flat_terrain.txt $P/phase_profile_flat_terrain_round_end.txt
cp gpurun_out/phase_flat_terrain_backlash.txt $P/phase_profile_backlash_round_end.txt
python3 - $P <<'PY'
import json, sys
P = sys.argv[1]
d = json.load(open(f"{P}/pmc_summary.json"))
f = d["FETCH_SIZE"]["mean"] * 1024; w = d["WRITE_SIZE"]["mean"] * 1024
t = json.load(open(f"{P}/traffic.json"))
t.update(hbm_bytes_per_launch=f + w, fetch=f, write=w)
json.dump(t, open(f"{P}/traffic.json", "w"))
print("traffic", f + w)
PY
