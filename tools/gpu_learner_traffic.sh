#!/bin/bash
# HBM traffic of ONE minibatch step of the learner, from counters:  tools/gpu_learner_traffic.sh TAG -> gpurun_out/ltraffic_TAG/learner_traffic.json
# Two separate --pmc passes (FETCH_SIZE; WRITE_SIZE -- they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") over the
# full-PPO loop (tools/gpu_train_bench.py flat_terrain_backlash 1), never combined with a trace domain.  Per learner kernel: mean bytes
# per launch and launches per minibatch step; the sum is what bench.py reports as the PPO legs' `roofline.traffic`.
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/ltraffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/gpu_train_bench.py flat_terrain_backlash 1 > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/gpu_train_bench.py flat_terrain_backlash 1 > /dev/null 2> $OUT/p2.err
python3 - $OUT $TAG <<'PY'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
LEARNER = ("mlp_fwd", "gae_kernel", "ppo_head_kernel", "ppo_gae_head", "mlp_bwd", "dw_gemm", "grad_finish", "adam_", "sqnorm")     # (gather_rows: the rollout's snapshot launches now, not the learner's)
WIDE = ("mlp_fwd", "mlp_bwd", "dw_gemm", "grad_finish")     # kernels whose reads are 16-byte-per-lane streams (Adam reads the torch-layout rows 4 bytes per lane)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = next((n for n in LEARNER if n in k), None)
        if name: agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
nadam = max(1, len(agg.get("adam_", {}).get("FETCH_SIZE", [])))          # one clip + Adam launch per minibatch step
res, total, total_raw = {}, 0.0, 0.0
for name in sorted(agg):
    f, w = agg[name].get("FETCH_SIZE", []), agg[name].get("WRITE_SIZE", [])
    if not f or not w: continue
    fetch, write = 1024 * sum(f) / len(f), 1024 * sum(w) / len(w)
    per_step = len(f) / nadam
    if name == "mlp_fwd" and per_step > 1.02:         # the rollout's policy launches share the kernel: keep the training launches (the larger ones)
        fs, ws = sorted(f)[-nadam:], sorted(w)[-nadam:]
        fetch, write, per_step = 1024 * sum(fs) / len(fs), 1024 * sum(ws) / len(ws), 1.0
    corr = 2.0 if name in WIDE else 1.0
    res[name] = {"launches_per_sgd_step": round(per_step, 3), "fetch_bytes_per_launch_raw": fetch, "fetch_correction": corr, "write_bytes_per_launch": write}
    total += per_step * (corr * fetch + write); total_raw += per_step * (fetch + write)
json.dump({"hbm_bytes_per_sgd_step": total, "hbm_bytes_per_sgd_step_uncorrected": total_raw, "kernels": res,
           "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/gpu_train_bench.py flat_terrain_backlash 1 (8192 envs, full PPO), tools/gpu_learner_traffic.sh {tag}: KB -> "
                   "bytes; FETCH_SIZE x 2 for the kernels that read 16-byte-per-lane streams (MI355X_MICROARCH.md: gfx950 tallies their 128-byte requests at 64 bytes), WRITE_SIZE as "
                   "reported; algorithmic activation traffic of a minibatch step: DESIGN.md 4.2"}, open(os.path.join(out, "learner_traffic.json"), "w"), indent=1)
print(open(os.path.join(out, "learner_traffic.json")).read())
PY
rm -rf $OUT/p1 $OUT/p2
