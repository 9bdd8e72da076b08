#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04u
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in 0 -1; do
rm -rf /tmp/p_small
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_small -- python3 $ROOT/bench.py --n 1000000 --batch 10000 --steps 5 --warmup 2 --force-type $t --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 --per-step-calls > $OUT/bench_small_t$t.json 2>/dev/null
python3 $ROOT/scripts/timeline.py /tmp/p_small 36 > $OUT/timeline_small_t$t.txt
done
cat $OUT/timeline_small_t-1.txt
echo done
