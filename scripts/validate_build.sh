#!/bin/bash
# GPU suite + smoke + the default bench with the driver's flags on a GPU box: gpurun -- bash scripts/validate_build.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/validate
mkdir -p $OUT
cd $ROOT
timeout -k 10 850 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1 || { tail -60 $OUT/tests_gpu.log; exit 1; }
tail -3 $OUT/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
tail -2 $OUT/smoke.log
S=$(date +%s)
python bench.py --steps 20 --warmup 5 > $OUT/bench_default_driver_flags.json 2>$OUT/bench_default.err || { tail -30 $OUT/bench_default.err; exit 1; }
echo "bench wall $(( $(date +%s) - S )) s"
python - <<PY
import json
o=json.load(open("$OUT/bench_default_driver_flags.json")); r=o["roofline"]
print("value %.0f q/s  ms/step %.1f  frac %.4f  retried %d  launches %d  cpu %.1f q/s on %d threads" % (o["value"], o["ms_per_step"], r["frac"], r["retry_queries"], r["launches"], o["cpu_baseline"]["value"], o["cpu_baseline"]["threads"]))
c=o["configs12"]
for k in ("config1_type0","config2_mixed"): print(k, round(c[k]["resident_ms"],3), round(c[k]["host_to_host_ms"],3))
PY
echo done
