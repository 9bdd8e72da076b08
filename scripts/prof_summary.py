"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals and the filter per level."""
import csv, glob, sys
d = sys.argv[1]
st = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(st)))[:9]:
    print(f"{r['Name'][:34]:34s} calls={r['Calls']:>4s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_ms={float(r['AverageNs'])/1e6:8.3f} pct={r['Percentage']}")
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(tr)) if 'hvs_k_filter' in r['Kernel_Name']]
print("filter launches (last step):", [round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, 3) for r in rows[-6:]])
