#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd $ROOT
python bench.py --force-dist --steps 4 --warmup 1 --cpu-seconds 0 --no-e2e --no-configs12 --in-library --in-library-devices 0,0 > $OUT/bench_force_dist_rehearsal.json 2>/dev/null
python - <<PY
import json
f=json.load(open("$OUT/bench_force_dist_rehearsal.json"))
print(f.get('in_library'))
PY
bash scripts/collect_profiles.sh stats pmc 2>&1 | tail -12
echo done
