#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04m
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "rotated_int8 or nonuniform_vector_laws" > $OUT/tests_rot.log 2>&1; rc=$?
grep -E "profile|passed|failed|Error|assert" $OUT/tests_rot.log | head -40
[ $rc -ne 0 ] && { tail -30 $OUT/tests_rot.log; exit 1; }
for p in 3 1; do
HVS_TRACE=1 timeout -k 10 300 python bench.py --profile $p --steps 2 --warmup 1 --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 > $OUT/bench_profile${p}_auto.json 2>$OUT/bench_profile${p}_auto.err || echo "profile $p failed"
grep "planner" $OUT/bench_profile${p}_auto.err | head -4
python - <<PY
import json
o=json.load(open("$OUT/bench_profile${p}_auto.json")); r=o["roofline"]
print("profile $p auto: %.0f q/s  frac %.3f  rescored/query %.0f  retried %d  exact fallback %d  engine %d" % (o["value"], r["frac"], r["rescored_pairs_per_query"], r["retry_queries"], r["fallback_queries"], o["config"]["engine"]))
PY
done
echo done
