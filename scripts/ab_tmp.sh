set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
run() { python bench.py --cpu-seconds 0 --steps 3 --warmup 1 "$@" 2>gpurun_out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'filter ms/step', round(r['kernel_ms_avg']*r['launches']/d['steps'],1), 'frac', round(r['frac'],3))"; }
echo mixed; run
echo t0 b262144; run --batch 262144 --force-type 0
echo cfg2; run --n 1000000 --batch 10000
