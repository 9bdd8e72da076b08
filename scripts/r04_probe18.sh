#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04q
mkdir -p $OUT
cd $ROOT
B="python bench.py --engine 1 --batch 16384 --force-type 0 --steps 3 --warmup 1 --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12"
$B > $OUT/exact_shipped.json 2>/dev/null
HVS_LIB=$ROOT/scripts/experiments/libhvs_exact_exp1.so $B > $OUT/exact_half_lds_reads.json 2>/dev/null
HVS_LIB=$ROOT/scripts/experiments/libhvs_exact_exp2.so $B > $OUT/exact_no_multiplies.json 2>/dev/null
python - <<PY
import json
for f in ("exact_shipped","exact_half_lds_reads","exact_no_multiplies"):
    o=json.load(open("$OUT/"+f+".json")); r=o["roofline"]
    print("%-22s %8.0f q/s  %7.1f ms/step  kernel ms avg %.2f  frac %.3f" % (f, o["value"], o["ms_per_step"], r["kernel_ms_avg"], r["frac"]))
PY
echo done
