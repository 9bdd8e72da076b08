"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals and the filter per level."""
import csv, glob, sys
d = sys.argv[1]
st = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(st)))[:9]:
    print(f"{r['Name'][:34]:34s} calls={r['Calls']:>4s} total_ms={float(r['TotalDurationNs'])/1e6:9.2f} avg_ms={float(r['AverageNs'])/1e6:8.3f} pct={r['Percentage']}")
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(tr)) if 'hvs_k_filter' in r['Kernel_Name']]
durs = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
print("filter launches, all (planner probe, warm-up step and timed steps), ms:", [round(d, 3) for d in durs])
# bench.py's roofline covers the timed steps only: each step = the batch's level launches + its retry batch's
import json, os
bj = os.path.join(os.path.dirname(os.path.abspath(d.rstrip('/'))), 'bench_under_rocprof.json')
for cand in (bj, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'prof', 'bench_under_rocprof.json')):
    if os.path.exists(cand):
        b = json.load(open(cand))
        nl = int(b['roofline']['launches'])
        timed = durs[-nl:]
        print("timed region: last %d launches, rocprofv3 average %.3f ms (sum %.1f ms); bench.py's HIP-event average of the same launches %.3f ms"
              % (nl, sum(timed) / nl, sum(timed), b['roofline']['kernel_ms_avg']))
        break
