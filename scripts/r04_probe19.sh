#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04r
mkdir -p $OUT
cd $ROOT
timeout -k 10 850 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1 || { tail -60 $OUT/tests_gpu.log; exit 1; }
tail -3 $OUT/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
tail -6 $OUT/smoke.log
echo done
