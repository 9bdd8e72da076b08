#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04p
mkdir -p $OUT
cd $ROOT
python bench.py --only-configs12 > $OUT/configs12_default.json 2>/dev/null
python scripts/show_configs12.py $OUT/configs12_default.json default
HVS_SPLIT_SMALL=1 python bench.py --only-configs12 > $OUT/configs12_split.json 2>/dev/null
python scripts/show_configs12.py $OUT/configs12_split.json split_two_lanes
echo done
