#!/bin/bash
# Sample shader clock and socket power (rocm-smi) while ONE lab kernel is held running: tells whether a throughput gap between
# two kernels is a clock (power management) gap or an issue-stall gap.  Usage on a GPU box: scripts/clock_probe.sh > out.txt
cd "$(dirname "$0")/.."
for k in bare16 bare32 loop16 pipe16 pipe16x1; do
    ./scripts/shape_lab.out hold $k &
    pid=$!
    sleep 1.5
    for i in 1 2 3 4; do
        /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '
        echo " [$k]"
        sleep 0.5
    done
    wait $pid
done
