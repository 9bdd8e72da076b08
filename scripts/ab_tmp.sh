set -e
run() { python bench.py --cpu-seconds 0 --steps 2 --warmup 1 --batch 262144 "$@" 2>gpurun_out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'kern_ms', round(r['kernel_ms_avg'],3), 'resc/q', round(r['rescored_pairs_per_query']))"; }
echo base t0; run --force-type 0
echo nosync t0; HVS_LIB=scripts/libhvs_nosync.so run --force-type 0
echo base t2; run --force-type 2
echo nosync t2; HVS_LIB=scripts/libhvs_nosync.so run --force-type 2
