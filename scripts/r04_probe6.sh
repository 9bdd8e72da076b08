#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04f
mkdir -p $OUT
cd $ROOT
for th in 4 8 16; do
echo "HVS_STAGE_THREADS=$th"
HVS_STAGE_THREADS=$th HVS_TRACE=1 python scripts/share_probe.py 500000 2>&1 | grep -v amdgpu.ids | tail -5
done
echo done
