#!/bin/bash
# tools/ab.sh <lib_a.so> <lib_b.so> [bench args...] -- runs bench.py alternately with two builds of the library on the same box
# (box-to-box and run-to-run variation is +-4 %, larger than most single optimisations) and prints the stage timings.
a=$1; b=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for lib in $a $b; do
    AMX_LIBRARY=$root/$lib python $root/bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages']; print('$lib', d['ms_per_step'], {k: s[k]['avg_ms'] for k in s if isinstance(s[k], dict) and 'avg_ms' in s[k]})"
  done
done
