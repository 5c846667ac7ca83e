#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_lib.sh <git rev of the OLD source file> <file under rasr_amd/csrc> [bench args...]
# Builds rasr_amd/librasr_amd_old.so HERE (before gpurun) from the in-tree objects with <file> taken from <rev>, then on the GPU box runs
#   bench.py <bench args> three times per library (AMX_LIBRARY) and prints ms_per_step -- boxes differ by a few per cent, a
# comparison across two gpurun calls cannot resolve a 2 % change (how the DCT and filter-bank variants of round 4 were judged).
# usage:  tools/ab_lib.sh HEAD mfcc.hip --workload mfcc --steps 20 --warmup 3        (build step, in the container)
#         gpurun -- 'bash tools/ab_lib.sh run --workload mfcc --steps 20 --warmup 3'  (measurement, on the box)
cd "$(dirname "$0")/.."
if [ "$1" != "run" ]; then
  rev=$1; f=$2; shift 2
  flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value"
  case $f in mfcc.hip|gmm_fused.hip|gammatone.hip) flags="$flags -fno-slp-vectorize";; esac
  mkdir -p /tmp/ab_old && git show $rev:rasr_amd/csrc/$f > rasr_amd/csrc/ab_old_tmp_$f || exit 1
  (cd rasr_amd/csrc && /opt/rocm/bin/hipcc $flags -c ab_old_tmp_$f -o /tmp/ab_old/old.o 2>/dev/null; rm -f ab_old_tmp_$f
   objs=$(ls build/*.o | grep -v "build/${f%.*}.o")
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librasr_amd_old.so $objs /tmp/ab_old/old.o -lz -ldl) && ls -la rasr_amd/librasr_amd_old.so
  exit 0
fi
shift
for rep in 1 2 3; do
  for lib in librasr_amd_old.so librasr_amd.so; do
    AMX_LIBRARY=$PWD/rasr_amd/$lib python bench.py "$@" --no-cpu-baseline --no-configs 2>&1 | grep "^{" | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['ms_per_step'], {k: v.get('avg_ms') for k, v in d.get('stages', {}).items() if isinstance(v, dict)})"
  done
done
