#!/bin/bash
# rocprofv3 kernel statistics of the tied-model bench (run on the GPU box): tools/prof_tied.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tied
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tied -- python $GRAFT_REPO_ROOT/bench.py --workload gmm-tied --no-cpu-baseline --steps 20 "$@" 2>/dev/null | cut -c1-200
f=$(find /tmp/prof_tied -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-60s calls %4s avg %9.1f us  %5s %%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
else echo "no stats file"; find /tmp/prof_tied | head; fi
