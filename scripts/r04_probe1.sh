#!/bin/bash
# round 4, first GPU call: new tests, baseline of BASELINE configs[1]/[2] (resident / host->host), host trace, timelines
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "format_change or one_context or auto_engine" > $OUT/tests_new.log 2>&1 || { tail -40 $OUT/tests_new.log; exit 1; }
tail -3 $OUT/tests_new.log
HVS_TRACE=1 python bench.py --only-configs12 > $OUT/configs12_base.json 2> $OUT/configs12_trace.txt
cat $OUT/configs12_base.json
HVS_TRACE=1 python scripts/cli_e2e.py 1000000 10000 > $OUT/cli_e2e_1e6.txt 2>&1 || true
cat $OUT/cli_e2e_1e6.txt
cd /tmp && export TMPDIR=/tmp
for t in 0 -1; do
rm -rf /tmp/p_small
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_small -- python3 $ROOT/bench.py --n 1000000 --batch 10000 --steps 5 --warmup 2 --force-type $t --cpu-seconds 0 --no-e2e --no-fixed-q --no-configs12 > $OUT/bench_small_t$t.json 2>/dev/null
python3 $ROOT/scripts/timeline.py /tmp/p_small 50 > $OUT/timeline_small_t$t.txt
done
echo done
