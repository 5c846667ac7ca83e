#!/bin/bash
# A/B of two builds on one box: tools/build/librasr_amd_old.so ("alt") against the in-tree library over the bench workloads
run() { python bench.py "$@" --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['ms_per_step'], {k:v.get('avg_ms') for k,v in d.get('stages',{}).items() if isinstance(v,dict)})"; }
for w in "--workload mfcc --front-end mfplp" "--workload gmm --gmm-type batch-diagonal-maximum-float" "--workload gmm --gmm-type SIMD-diagonal-maximum" "--workload nn" "--workload gmm-tied" "--workload gmm-train --estimation-mode baum-welch" "--steps 4 --warmup 1"; do
  echo "== $w"
  for lib in tree alt tree alt; do
    if [ $lib = alt ]; then export AMX_LIBRARY=$GRAFT_REPO_ROOT/tools/build/librasr_amd_old.so; else unset AMX_LIBRARY; fi
    echo -n "$lib"; run $w
  done
done
