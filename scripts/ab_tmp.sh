set -e
run() { python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value']), 'eng', d['config']['engine'], 'kern_ms', round(r['kernel_ms_avg'],3), 'ach', round(r['achieved'],1), 'resc/q', round(r['rescored_pairs_per_query'],1), 'fb', r['fallback_queries'], 'dev_ms', round(r['device_query_ms_per_step'],1), 'load_s', round(d['load_s'],2))"; }
for f in bf16 i8; do
echo $f mixed; HVS_FILTER_FORMAT=$f run --batch 262144
echo $f t0; HVS_FILTER_FORMAT=$f run --batch 65536 --force-type 0
done
echo auto mixed; run --batch 262144
