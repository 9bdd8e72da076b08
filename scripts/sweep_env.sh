#!/bin/bash
# A/B sweep of run-time knobs: one bench.py line per setting ("VAR=val VAR2=val2" per argument; "-" = defaults).
# Usage: scripts/sweep_env.sh out.txt "<bench args>" "HVS_GUESS_MID=8" "HVS_RADIX_LAST=8 HVS_GUESS_MID=8" ...
out=$1; shift; bargs=$1; shift
: > "$out"
for setting in "$@"; do
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  line=$(env $envs python bench.py --cpu-seconds 0 --no-e2e $bargs 2>/dev/null | tail -1)
  echo "$setting :: $(python - "$line" <<'PY'
import json,sys
d=json.loads(sys.argv[1]); r=d["roofline"]
print("q/s %.0f ms/step %.1f frac %.4f filter_ms/step %.1f launches %d rescored/q %.0f retry %d fallback %d" % (d["value"], d["ms_per_step"], r["frac"], r["kernel_ms_avg"]*r["launches"]/d["steps"], r["launches"], r["rescored_pairs_per_query"], r["retry_queries"], r["fallback_queries"]))
PY
)" | tee -a "$out"
done
