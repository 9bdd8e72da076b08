#!/bin/bash
# round 4, second GPU call: whole GPU suite on the merged prep kernel, then configs[1]/[2] sweeps
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04b
mkdir -p $OUT
cd $ROOT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1 || { tail -60 $OUT/tests_gpu.log; exit 1; }
tail -3 $OUT/tests_gpu.log
python bench.py --only-configs12 > $OUT/configs12_default.json 2>/dev/null
python scripts/show_configs12.py $OUT/configs12_default.json default
for pf in 4 6 7; do
HVS_GUESS_PFAIL=$pf python bench.py --only-configs12 > $OUT/configs12_pfail$pf.json 2>/dev/null
python scripts/show_configs12.py $OUT/configs12_pfail$pf.json pfail$pf
done
for si in 2 8 16; do
HVS_SEG_ITEMS=$si python bench.py --only-configs12 > $OUT/configs12_segitems$si.json 2>/dev/null
python scripts/show_configs12.py $OUT/configs12_segitems$si.json segitems$si
done
HVS_TRACE=1 python scripts/cli_e2e.py 1000000 10000 > $OUT/cli_e2e_1e6.txt 2>&1 || true
cat $OUT/cli_e2e_1e6.txt
echo done
