#!/bin/bash
# rocprofv3 kernel statistics of the tied-model bench (run on the GPU box): tools/prof_tied.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tied
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tied -- python $GRAFT_REPO_ROOT/bench.py --workload gmm-tied --no-cpu-baseline --steps 20 "$@" 2>/dev/null | cut -c1-200
f=$(find /tmp/prof_tied -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -9 "$f" | cut -c1-170; else echo "no stats file"; find /tmp/prof_tied | head; fi
