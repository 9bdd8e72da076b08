#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04o
mkdir -p $OUT
cd $ROOT
timeout -k 10 420 python scripts/fuzz_engines.py 41 300 > $OUT/fuzz_default.txt 2>&1; echo "rc $?"; tail -3 $OUT/fuzz_default.txt
HVS_MFMA_BATCH=8192 timeout -k 10 420 python scripts/fuzz_engines.py 42 300 > $OUT/fuzz_small_batches_two_lanes.txt 2>&1; echo "rc $?"; tail -3 $OUT/fuzz_small_batches_two_lanes.txt
grep -c " ok" $OUT/fuzz_default.txt $OUT/fuzz_small_batches_two_lanes.txt
grep -h "MISMATCH" $OUT/*.txt | head
echo done
