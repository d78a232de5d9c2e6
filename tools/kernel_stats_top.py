"""Prints the top rows of a rocprofv3 kernel_stats.csv found under DIR:  python tools/kernel_stats_top.py DIR [N]"""
import csv, glob, os, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12
fs = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if not fs:
    print("no kernel_stats.csv under", d, os.listdir(d) if os.path.isdir(d) else "(missing)"); sys.exit(0)
for r in list(csv.DictReader(open(fs[0])))[:n]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} tot_ms={float(r['TotalDurationNs'])/1e6:9.2f}")
