#!/bin/bash
# ablations of gemm_mx_kernel (lab build of the library: make -C rasr_amd/csrc OBJDIR=build_lab OUT=../../tools/build/librasr_amd_lab.so EXTRA=-DAMX_LAB)
# usage: tools/ab_mx.sh [dbg values...]   prints the output-layer time (ffnn_gemm_max) and the mean GEMM time of the NN leg
cd "$(dirname "$0")/.."
export AMX_LIBRARY=$PWD/tools/build/librasr_amd_lab.so
for d in "${@:-0 8 16 24 32 64 72}"; do
  for dd in $d; do
    AMX_TUNING=mx_dbg=$dd python bench.py --workload nn-pipeline --precision f16mx --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>&1 | grep "^{" | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); s=d['stages']; print('dbg %3s  step %.3f ms  output layer %.3f ms  mean gemm %.3f ms' % ('$dd', d['ms_per_step'], s['ffnn_gemm_max']['avg_ms'], s['ffnn_gemm']['avg_ms']))"
  done
done
