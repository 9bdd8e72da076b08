#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04g
mkdir -p $OUT
cd $ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -s -k "d1e8_config4" > $OUT/test_config4.log 2>&1 || { tail -40 $OUT/test_config4.log; exit 1; }
grep -E "D=1e8|passed|failed" $OUT/test_config4.log
S=$(date +%s)
python bench.py --steps 20 --warmup 5 > $OUT/bench_default_driver_flags.json 2>$OUT/bench_default.err || { tail -30 $OUT/bench_default.err; exit 1; }
echo "bench wall $(( $(date +%s) - S )) s"
python - <<PY
import json
o=json.load(open("$OUT/bench_default_driver_flags.json")); r=o["roofline"]
print("value %.0f q/s  ms/step %.1f  frac %.4f  device ms/step %.1f  retried %d  launches %d" % (o["value"], o["ms_per_step"], r["frac"], r["device_query_ms_per_step"], r["retry_queries"], r["launches"]))
print("e2e", o["end_to_end"]["value"], "cpu", json.dumps(o["cpu_baseline"])[:900])
print("recall", o["recall_at_100"], o["recall_checked_queries"], o["parity"])
for s in o["fixed_q"]["shares"]: print(s["n_gpus"], s["queries"], "%.0f"%s["resident"]["value"], "%.0f"%s["host_to_host"]["value"])
print(json.dumps(o["configs12"])[:1500])
PY
echo done
