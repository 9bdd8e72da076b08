#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04w
mkdir -p $OUT
cd $ROOT
python bench.py --force-dist --steps 4 --warmup 1 --cpu-seconds 0 --no-e2e --no-configs12 --in-library --in-library-devices 0,0 > $OUT/bench_force_dist.json 2>$OUT/bench_force_dist.err || { tail -30 $OUT/bench_force_dist.err; exit 1; }
python - <<PY
import json
o=json.load(open("$OUT/bench_force_dist.json"))
print("value", o["value"], o["config"]["timed_region"])
print("collective", o.get("collective"))
print("fixed_q", {k:v for k,v in o.get("fixed_q",{}).items() if k!="scope"})
print("in_library", o.get("in_library"))
PY
python bench.py --in-library-only --in-library-devices 0 2>/dev/null | tail -1
echo done
