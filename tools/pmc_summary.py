"""Summarises rocprofv3 --pmc counter CSVs (one or more passes) for the step kernel into one JSON:
    python tools/pmc_summary.py OUT.json DIR [DIR ...]
Per counter: mean / min / max over the dispatches of kernels whose name contains `step_kernel`."""
import csv, glob, json, os, sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
vals = defaultdict(list)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "step_kernel" in row.get("Kernel_Name", ""):
                    vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in sorted(vals.items())}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: round(v["mean"], 1) for k, v in res.items()}))
