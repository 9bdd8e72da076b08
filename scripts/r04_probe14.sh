#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04n
mkdir -p $OUT
cd $ROOT
timeout -k 10 850 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1 || { tail -60 $OUT/tests_gpu.log; exit 1; }
tail -3 $OUT/tests_gpu.log
S=$(date +%s)
python bench.py --steps 20 --warmup 5 > $OUT/bench_default_driver_flags.json 2>$OUT/bench_default.err || { tail -30 $OUT/bench_default.err; exit 1; }
echo "bench wall $(( $(date +%s) - S )) s"
python - <<PY
import json
o=json.load(open("$OUT/bench_default_driver_flags.json")); r=o["roofline"]
print("value %.0f q/s  ms/step %.1f  frac %.4f  device ms/step %.1f  retried %d  launches %d" % (o["value"], o["ms_per_step"], r["frac"], r["device_query_ms_per_step"], r["retry_queries"], r["launches"]))
print("e2e", o["end_to_end"]["value"], "cpu", json.dumps(o["cpu_baseline"])[:600])
print("recall", o["recall_at_100"], o["recall_checked_queries"], o["parity"])
for s in o["fixed_q"]["shares"]: print(s["n_gpus"], s["queries"], "%.0f"%s["resident"]["value"], "%.0f"%s["host_to_host"]["value"])
c=o["configs12"]
for k in ("config1_type0","config2_mixed"): print(k, {a: round(b,3) if isinstance(b,float) else b for a,b in c[k].items()})
print("load", c["load_data_ms"], c["load_data_ms_second_call"])
PY
echo done
