#!/bin/bash
# Copies the evidence gathered on the GPU box (gpurun_out/, scratch) into profiles/<round>/ (tracked):
#   tools/collect_round_profiles.sh r4      after tools/gpu_profile_round.sh r4 (kernel traces, counters, bench lines), and -- when they ran --
#   tools/gpu_learner_trace.sh r4end, tools/gpu_train_runs.sh r4end, tools/hfield_variants.py, a `pytest -m gpu` run (parity_worst.json)
set -eu   # a missing REQUIRED input must fail loudly: stale evidence in profiles/ is worse than none
R=${1:-r5}; P=profiles/$R; G=gpurun_out/prof_$R
mkdir -p $P
for t in "" _flat_terrain_backlash _rough_terrain_backlash; do
  cp $G/kernel_stats$t.csv $G/bench_under_rocprof$t.json $G/pmc_summary$t.json $G/traffic$t.json $P/
done
cp $G/bench_default_run.json $G/bench_flat_terrain_backlash.json $G/bench_rough_terrain_backlash.json $G/bench_ppo_config3.json $G/bench_ppo_config4.json $G/bench_ppo_flat.json $P/
opt() { if [ -e "$1" ]; then cp "$1" "$2"; else echo "(optional input $1 absent)"; fi; }
opt gpurun_out/ltrace_${R}end/timeline.txt $P/learner_step_timeline.txt
opt gpurun_out/ltrace_${R}end/stats.txt $P/learner_kernel_stats_top.txt
opt gpurun_out/parity_worst.json $P/parity_worst.json
opt $G/phase_profile_rough_terrain.txt $P/phase_profile_rough_terrain.txt   # ODK_LIB=.../libodk_prof.so python3 tools/gpu_phase_profile.py rough_terrain_backlash
opt $G/phase_profile_flat_terrain.txt $P/phase_profile_flat_terrain.txt
opt $G/hf_run_twice.txt $P/hf_run_twice.txt                                   # tools/gpu_hf_knock.sh 0 256 512 1024 2048 4096 8192 16384 0
for t in flat backlash rough standing rough_up_normals; do opt gpurun_out/train_${R}end/$t/metrics.jsonl $P/train_${t}_metrics.jsonl; done
opt gpurun_out/train_${R}end/wall.txt $P/train_wall_times.txt
ls $P
