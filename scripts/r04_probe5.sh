#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04e
mkdir -p $OUT
cd $ROOT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1 || { tail -60 $OUT/tests_gpu.log; exit 1; }
tail -3 $OUT/tests_gpu.log
python bench.py --force-dist --steps 4 --warmup 1 --cpu-seconds 0 --no-e2e --no-configs12 --in-library --in-library-devices 0,0 > $OUT/bench_force_dist.json 2>$OUT/bench_force_dist.err || { tail -30 $OUT/bench_force_dist.err; exit 1; }
python - <<PY
import json
o=json.load(open("$OUT/bench_force_dist.json"))
print("value", o["value"], "ms/step", o["ms_per_step"], o["config"]["timed_region"])
print("collective", o.get("collective"))
print("fixed_q", o.get("fixed_q"))
print("in_library", o.get("in_library"))
PY
echo done
