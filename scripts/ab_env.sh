#!/bin/bash
# A/B on one GPU box by environment: scripts/ab_env.sh "VAR=1" "VAR=0 OTHER=2" ... -- [bench args]; each configuration runs
# bench.py's resident leg once; the first configuration is repeated at the end (drift check)
cd "$(dirname "$0")/.."
cfgs=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; extra=("$@"); break; fi; cfgs+=("$1"); shift; done
one() {
    env $1 python bench.py --no-e2e --cpu-seconds 0 "${extra[@]}" 2>gpurun_out/ab_err.txt | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('[$1]', 'value', round(j['value']), 'ms', round(j['ms_per_step'],1), 'frac', round(r['frac'],4), 'kernel_ms_avg', round(r['kernel_ms_avg'],2), 'launches', r['launches'])"
}
for c in "${cfgs[@]}"; do one "$c" || { cat gpurun_out/ab_err.txt | tail -5; exit 1; }; done
one "${cfgs[0]}"
