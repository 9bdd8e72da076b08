#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04k
mkdir -p $OUT
cd $ROOT
for b in 262144 2097152; do
HVS_DEMOTE=0 timeout -k 10 300 python bench.py --profile 3 --batch $b --steps 2 --warmup 1 --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 > $OUT/bench_profile3_nodemote_$b.json 2>/dev/null || echo "failed"
python - <<PY
import json
o=json.load(open("$OUT/bench_profile3_nodemote_$b.json")); r=o["roofline"]
print("profile 3 auto, no demote, batch $b: %.0f q/s  frac %.3f  rescored/query %.0f  retried %d  exact fallback %d  engine %d" % (o["value"], r["frac"], r["rescored_pairs_per_query"], r["retry_queries"], r["fallback_queries"], o["config"]["engine"]))
PY
done
echo done
