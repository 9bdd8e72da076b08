#!/bin/bash
# Samples rocm-smi power/clock while bench.py runs (GPU box): is the filter power-limited?
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --cpu-seconds 0 --steps 30 --warmup 1 > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 5
for i in $(seq 1 40); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks -d 0 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo
  sleep 0.2
done
wait $BP
echo idle:
/opt/rocm/bin/rocm-smi --showpower --showclocks --showmaxpower -d 0 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo
