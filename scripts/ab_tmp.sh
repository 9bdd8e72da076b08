set -e
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_m
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > /tmp/b.json 2>/dev/null
python3 $GRAFT_REPO_ROOT/scripts/prof_summary.py /tmp/prof_m | head -6
cd $GRAFT_REPO_ROOT
python bench.py --cpu-seconds 0 --steps 3 --warmup 2 --n 1000000 --batch 10000 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('cfg2', d['value'], d['ms_per_step'])"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "levels_and_ranges or ragged or goldens" 2>&1 | tail -2
