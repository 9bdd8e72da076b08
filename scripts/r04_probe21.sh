#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04t
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "exact or ragged or each_query_type or ties or fp_known or many_small or baseline_engine or k_sweep or other_k" > $OUT/tests_exact.log 2>&1; rc=$?
tail -3 $OUT/tests_exact.log
[ $rc -ne 0 ] && { tail -40 $OUT/tests_exact.log; exit 1; }
B="python bench.py --engine 1 --batch 16384 --steps 3 --warmup 1 --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12"
$B --force-type 0 > $OUT/exact_type0.json 2>/dev/null
$B > $OUT/exact_mixed.json 2>/dev/null
python - <<PY
import json
for f in ("exact_type0","exact_mixed"):
    o=json.load(open("$OUT/"+f+".json")); r=o["roofline"]
    print("%-22s %8.0f q/s  %7.1f ms/step  kernel ms avg %.2f  frac %.3f" % (f, o["value"], o["ms_per_step"], r["kernel_ms_avg"], r["frac"]))
PY
echo done
