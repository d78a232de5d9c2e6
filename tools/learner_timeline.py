"""Timeline of ONE learner minibatch step from a rocprofv3 kernel trace (tools/gpu_train_bench.py under --kernel-trace):
    python tools/learner_timeline.py DIR [step_index]
Prints every kernel of the chosen step (between two adam_kernel launches): start offset, duration, queue, name."""
import csv, glob, os, sys
d = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 300
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_" in r["Kernel_Name"] and "sqnorm" not in r["Kernel_Name"]]
a, b = adam[which], adam[which + 1]
t0 = int(rows[a]["End_Timestamp"])
busy = 0
for r in rows[a + 1: b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    busy += e - s
    print(f"{s/1e3:8.1f} us  +{(e-s)/1e3:6.1f}  q{r.get('Queue_Id','?'):>3s}  grid={r.get('Grid_Size','?'):>8s}  {r['Kernel_Name'][:80]}")
print(f"step wall {(int(rows[b]['End_Timestamp']) - t0)/1e3:.1f} us, sum of kernel durations {busy/1e3:.1f} us")
