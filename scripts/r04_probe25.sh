#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04x
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python scripts/fuzz_engines.py 44 420 > $OUT/fuzz_seed44.txt 2>&1; echo "fuzz rc $?"; tail -1 $OUT/fuzz_seed44.txt
HVS_MFMA_BATCH=2048 timeout -k 10 500 python scripts/fuzz_engines.py 45 420 > $OUT/fuzz_seed45_small_batches.txt 2>&1; echo "fuzz rc $?"; tail -1 $OUT/fuzz_seed45_small_batches.txt
grep -h MISMATCH $OUT/*.txt | head -3
echo done
