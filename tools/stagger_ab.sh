cd /root/repo
export AMX_LIBRARY=$PWD/rasr_amd/librasr_amd_lab.so
mkdir -p gpurun_out/r04
for rep in 1 2; do
for s in 0 100 300 600; do
  AMX_TUNING=stagger=$s python bench.py --workload nn-pipeline --precision f16mx --steps 8 --warmup 2 --no-cpu-baseline --no-configs 2>&1 | grep "^{" | tail -1 | \
   python -c "import sys,json; d=json.loads(sys.stdin.readline()); s=d['stages']; print('stagger %4s  step %.3f ms  output layer %.3f ms  mean gemm %.3f ms' % ('$s', d['ms_per_step'], s['ffnn_gemm_max']['avg_ms'], s['ffnn_gemm']['avg_ms']))"
done; done | tee gpurun_out/r04/stagger2.log
