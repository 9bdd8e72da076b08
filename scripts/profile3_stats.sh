#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04y
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_p3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_p3 -- python3 $ROOT/bench.py --profile 3 --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 > $OUT/bench_profile3_under_rocprof.json 2>/dev/null
python3 $ROOT/scripts/prof_summary.py /tmp/p_p3 > $OUT/summary_profile3.txt
python3 $ROOT/scripts/timeline.py /tmp/p_p3 60 > $OUT/timeline_profile3.txt
cat $OUT/summary_profile3.txt | head -14
grep -v "fillBuffer\|ROCPRIM" $OUT/timeline_profile3.txt | tail -42 | head -30
