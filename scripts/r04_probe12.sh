#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04l
mkdir -p $OUT
cd $ROOT
run() {
  tag=$1; shift
  env "$@" HVS_TRACE=1 HVS_DEMOTE=0 timeout -k 10 300 python bench.py --profile 3 --batch 262144 --steps 1 --warmup 1 --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 > $OUT/b_$tag.json 2>$OUT/b_$tag.err
  grep "planner\|tiles built" $OUT/b_$tag.err | head -8
  python - <<PY
import json
o=json.load(open("$OUT/b_$tag.json")); r=o["roofline"]
print("$tag: %.0f q/s  rescored/query %.0f  retried %d  exact fallback %d  engine %d" % (o["value"], r["rescored_pairs_per_query"], r["retry_queries"], r["fallback_queries"], o["config"]["engine"]))
PY
}
run auto X=1
run forced_i8 HVS_FILTER_FORMAT=i8
run auto_noprobe HVS_PLAN_PROBE=0
echo done
