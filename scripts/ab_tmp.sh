set -e
run() { python bench.py --cpu-seconds 0 --steps 3 --warmup 1 "$@" 2>gpurun_out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'filter ms/step', round(r['kernel_ms_avg']*r['launches']/d['steps'],1), 'resc/q', round(r['rescored_pairs_per_query'],1))"; }
echo b262144; run
echo b524288; HVS_MFMA_BATCH=524288 run --batch 524288
echo b1048576; HVS_MFMA_BATCH=1048576 run --batch 1048576 --steps 2
echo b131072; run --batch 131072
