set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "levels_and_ranges or ragged or overflow or goldens or clustered" 2>&1 | tail -2
run() { python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" 2>gpurun_out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value']), 'eng', d['config']['engine'], 'kern_ms', round(r['kernel_ms_avg'],3), 'launches', r['launches'], 'ach', round(r['achieved'],1), 'resc/q', round(r['rescored_pairs_per_query'],1), 'fb', r['fallback_queries'], 'dev_ms', round(r['device_query_ms_per_step'],1))"; }
echo mixed; run --batch 262144
echo t0; run --batch 65536 --force-type 0
echo t2; run --batch 65536 --force-type 2
echo mixed bf16; HVS_FILTER_FORMAT=bf16 run --batch 262144
