#!/bin/bash
# A/B of two library builds on the NN leg: tools/ab_nn.sh <lib.so>...  ("" = in-tree)
for rep in 1 2; do
for l in "" "$@"; do
  if [ -z "$l" ]; then unset AMX_LIBRARY; else export AMX_LIBRARY=$GRAFT_REPO_ROOT/$l; fi
  echo -n "${l:-in-tree}: "; python bench.py --workload nn-pipeline --steps 10 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
done
