#!/bin/bash
# A/B of library builds: one bench.py line per (library, bench arguments) pair.
# Usage: scripts/ab_libs.sh out.txt "lib1.so lib2.so ..." "<bench args 1>" "<bench args 2>" ...
out=$1; shift; libs=$1; shift
: > "$out"
for L in $libs; do
  for a in "$@"; do
    line=$(HVS_LIB=$PWD/$L python bench.py --cpu-seconds 0 --no-e2e --no-fixed-q $a 2>/dev/null | tail -1)
    echo "$L $a :: $(python - "$line" <<'PY'
import json,sys
d=json.loads(sys.argv[1]); r=d["roofline"]
print("q/s %.0f ms/step %.1f frac %.4f filter_ms/step %.1f kernel %s rescored/q %.0f" % (d["value"], d["ms_per_step"], r["frac"], r["kernel_ms_avg"]*r["launches"]/d["steps"], r["kernel"], r["rescored_pairs_per_query"]))
PY
)" | tee -a "$out"
  done
done
