set -e
run() { python bench.py --cpu-seconds 0 --steps 3 --warmup 1 "$@" 2>gpurun_out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'kern_ms', round(r['kernel_ms_avg'],3))"; }
echo base; run
echo stg10; HVS_LIB=scripts/libhvs_stg10.so run
HVS_LIB=scripts/libhvs_stg10.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "levels_and_ranges or ragged" 2>&1 | tail -2
