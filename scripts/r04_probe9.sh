#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04i
mkdir -p $OUT
cd $ROOT
HVS_TRACE=1 python scripts/cli_e2e.py 1000000 10000 > $OUT/cli_e2e_1e6.txt 2>&1 || true
cat $OUT/cli_e2e_1e6.txt
for p in 2 3 4; do
HVS_TRACE=1 timeout -k 10 240 python bench.py --profile $p --batch 262144 --steps 1 --warmup 1 --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 > $OUT/bench_profile${p}_auto.json 2>$OUT/bench_profile${p}_auto.err || echo "profile $p failed"
grep "planner" $OUT/bench_profile${p}_auto.err | head -4
python - <<PY
import json
o=json.load(open("$OUT/bench_profile${p}_auto.json")); r=o["roofline"]
print("profile $p auto: %.0f q/s  frac %.3f  rescored/query %.0f  retried %d  exact fallback %d  engine %d" % (o["value"], r["frac"], r["rescored_pairs_per_query"], r["retry_queries"], r["fallback_queries"], o["config"]["engine"]))
PY
done
python bench.py --steps 2 --warmup 1 --no-e2e --no-fixed-q --no-configs12 --cpu-seconds 60 > $OUT/bench_cpu.json 2>/dev/null
python - <<PY
import json
o=json.load(open("$OUT/bench_cpu.json"))
print("cpu", json.dumps(o["cpu_baseline"])[:700])
PY
echo done
