"""Timeline of ONE rollout step from a rocprofv3 kernel trace (tools/gpu_train_bench.py under --kernel-trace):
    python tools/rollout_timeline.py DIR [step_index]
Prints every kernel between two consecutive env-step launches."""
import csv, glob, os, sys
d = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 45
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [i for i, r in enumerate(rows) if "step_kernel" in r["Kernel_Name"]]
a, b = st[which], st[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a: b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:8.1f} us  +{(e-s)/1e3:7.1f}  {r['Kernel_Name'][:110]}")
