"""Kernel timeline of the LAST step of a rocprofv3 --kernel-trace run: start, duration, queue and overlap with the previous kernel.
Usage: python scripts/timeline.py <rocprof output dir> [n_last_kernels]"""
import csv, glob, sys
d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 120
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(tr)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the run ends with the download of the first timed batch (a tail of copy kernels): the last step ends at the last hvs kernel
last = max(i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(('hvs_k', 'void hvs_k')))
rows = rows[max(0, last + 1 - n_last):last + 1]
t0 = int(rows[0]['Start_Timestamp'])
prev_end = t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    ov = max(0, min(e, prev_end) - s)
    print(f"{r['Kernel_Name'][:28]:28s} q{r.get('Queue_Id','?'):>3s} start {(s-t0)/1e6:9.3f} ms dur {(e-s)/1e6:8.3f} ms  under earlier kernels {ov/1e6:8.3f} ms")
    prev_end = max(prev_end, e)
