#!/bin/bash
# round 4, third GPU call: GPU suite on two lanes, then the default bench with / without lanes
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04c
mkdir -p $OUT
cd $ROOT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1 || { tail -60 $OUT/tests_gpu.log; exit 1; }
tail -3 $OUT/tests_gpu.log
B="python bench.py --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 --steps 8 --warmup 2"
$B > $OUT/bench_lanes1.json 2>$OUT/bench_lanes1.err || { tail -20 $OUT/bench_lanes1.err; exit 1; }
HVS_LANES=0 $B > $OUT/bench_lanes0.json 2>/dev/null
$B --per-step-calls > $OUT/bench_per_step_calls.json 2>/dev/null
python - <<PY
import json
for f in ("bench_lanes1","bench_lanes0","bench_per_step_calls"):
    o=json.load(open("$OUT/"+f+".json")); r=o["roofline"]
    print(f, "value %.0f q/s  ms/step %.1f  frac %.4f  device ms/step %.1f  retried %d  launches %d" % (o["value"], o["ms_per_step"], r["frac"], r["device_query_ms_per_step"], r["retry_queries"], r["launches"]))
PY
echo done
