"""Median duration per (kernel, grid) of a rocprofv3 kernel trace found under DIR:  python tools/kernel_trace_medians.py DIR [min_count]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    key = (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    if len(v) >= int(sys.argv[2]) if len(sys.argv) > 2 else 100: print(f"{k[0]:70s} grid=({k[1]},{k[2]}) n={len(v)} median {sorted(v)[len(v)//2]/1e3:.1f} us")
