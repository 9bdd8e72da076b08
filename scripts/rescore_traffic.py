"""HBM-side traffic of the re-scoring launches, per level: rocprofv3 --pmc FETCH_SIZE pass -> bytes per launch next to the
algorithmic bytes (rows re-scored x 408 B is not known per launch here; the bench line's rescored_pairs_per_query / levels gives the mean).
Usage: python scripts/rescore_traffic.py <rocprof dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = {}
for r in csv.DictReader(open(f)):
    if "hvs_k_rescore" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
vals = [per[k] for k in sorted(per)]
print("re-score launches:", len(vals))
print("fetched GB per launch (FETCH_SIZE KiB x 2 x 1024):", [round(2.0 * v * 1024.0 / 1e9, 2) for v in vals])
