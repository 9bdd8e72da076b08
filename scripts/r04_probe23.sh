#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04v
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "two_lanes" > $OUT/tests_lanes.log 2>&1; rc=$?
grep -E "engine|passed|failed|Error" $OUT/tests_lanes.log | tail -20
[ $rc -ne 0 ] && tail -30 $OUT/tests_lanes.log
timeout -k 10 700 python scripts/fuzz_engines.py 43 560 > $OUT/fuzz_seed43.txt 2>&1; echo "fuzz rc $?"; tail -2 $OUT/fuzz_seed43.txt
echo done
