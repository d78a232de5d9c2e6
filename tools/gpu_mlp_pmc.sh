#!/bin/bash
# PMC passes over the learner's network kernels (tools/gpu_mlp_bench.py): tools/gpu_mlp_pmc.sh TAG -> gpurun_out/mlp_pmc_TAG/*.csv summaries
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/mlp_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/gpu_mlp_bench.py > /dev/null 2> $OUT/p$i.err
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "fwd" if "mlp_fwd" in k else "bwd" if "mlp_bwd" in k else "dw" if "dw_gemm" in k else None
        if name: agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for name in sorted(agg):
        for c in sorted(agg[name]):
            v = agg[name][c]
            line = f"{name:4s} {c:40s} n={len(v):4d} median={sorted(v)[len(v)//2]:.4g} max={max(v):.4g}"
            print(line); fo.write(line + "\n")
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
