#!/bin/bash
# A/B on one GPU box: bench.py (resident leg only) with the default library and with each alternative build given as argument,
# default first and last (drift check).  Usage: scripts/ab_bench.sh scripts/libhvs_x.so ... [-- extra bench args]
cd "$(dirname "$0")/.."
libs=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; extra=("$@"); break; fi; libs+=("$1"); shift; done
one() {  # label, lib
    if [ -n "$2" ]; then export HVS_LIB="$2"; else unset HVS_LIB; fi
    python bench.py --no-e2e --cpu-seconds 0 "${extra[@]}" 2>gpurun_out/ab_err.txt | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('$1', 'value', j['value'], 'ms', j['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r.get('kernel_ms_per_launch'))"
}
one default "" || exit 1
for l in "${libs[@]}"; do one "$l" "$l" || exit 1; done
one default ""
