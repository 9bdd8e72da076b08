#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $ROOT
B="python bench.py --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 --steps 8 --warmup 2"
$B > $OUT/bench_lanes1.json 2>$OUT/bench_lanes1.err || { tail -20 $OUT/bench_lanes1.err; exit 1; }
HVS_HEAD_PRIORITY=0 $B > $OUT/bench_lanes0.json 2>/dev/null
python - <<PY
import json
for f in ("bench_lanes1","bench_lanes0"):
    o=json.load(open("$OUT/"+f+".json")); r=o["roofline"]
    print(f, "value %.0f q/s  ms/step %.1f  frac %.4f  device ms/step %.1f  retried %d  launches %d" % (o["value"], o["ms_per_step"], r["frac"], r["device_query_ms_per_step"], r["retry_queries"], r["launches"]))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_l
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_l -- python3 $ROOT/bench.py --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2>/dev/null
python3 $ROOT/scripts/timeline.py /tmp/p_l 140 > $OUT/timeline_lanes.txt
echo done
