#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $ROOT
HVS_TRACE=1 python scripts/cli_e2e.py 1000000 10000 > $OUT/cli_e2e_1e6.txt 2>&1 || true
cat $OUT/cli_e2e_1e6.txt
S=$(date +%s)
python bench.py --steps 4 --warmup 1 --no-e2e > $OUT/bench_short.json 2>$OUT/bench_short.err || { tail -30 $OUT/bench_short.err; exit 1; }
echo "bench wall $(( $(date +%s) - S )) s"
python - <<PY
import json
o=json.load(open("$OUT/bench_short.json")); r=o["roofline"]
print("value %.0f q/s  ms/step %.1f  frac %.4f" % (o["value"], o["ms_per_step"], r["frac"]))
print("cpu", json.dumps(o["cpu_baseline"])[:1200])
for s in o["fixed_q"]["shares"]: print(s["n_gpus"], s["queries"], "%.0f"%s["resident"]["value"], "%.0f"%s["host_to_host"]["value"])
PY
for p in 2 3 4; do
for f in i8 f16; do
HVS_FILTER_FORMAT=$f HVS_DEMOTE=0 timeout -k 10 240 python bench.py --profile $p --batch 262144 --steps 1 --warmup 1 --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 > $OUT/bench_profile${p}_$f.json 2>/dev/null || echo "profile $p $f failed/timeout"
python - <<PY
import json
try:
    o=json.load(open("$OUT/bench_profile${p}_$f.json")); r=o["roofline"]
    print("profile $p $f: %.0f q/s  frac %.3f  rescored/query %.0f  retried %d  exact fallback %d  engine %d" % (o["value"], r["frac"], r["rescored_pairs_per_query"], r["retry_queries"], r["fallback_queries"], o["config"]["engine"]))
except Exception as e: print("profile $p $f:", e)
PY
done
done
echo done
