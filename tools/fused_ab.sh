#!/bin/bash
# A/B of gmm_fused_kernel builds on one box: tools/fused_ab.sh <lib.so>... (paths relative to the repo; "" = the in-tree library)
for rep in 1 2; do
for l in "" "$@"; do
  if [ -z "$l" ]; then unset AMX_LIBRARY; else export AMX_LIBRARY=$GRAFT_REPO_ROOT/$l; fi
  echo -n "${l:-in-tree}: "; python bench.py --workload gmm-train --steps 5 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['stages']; print(d['ms_per_step'], {k:s[k].get('avg_ms') for k in s if isinstance(s[k],dict)})"
done
done
