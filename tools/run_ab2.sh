cd /root/repo
python tools/gmm_store_ab.py 2>&1 | grep -v amdgpu.ids | (mkdir -p gpurun_out/r04; tee gpurun_out/r04/gmm_store_ab.log)
export AMX_LIBRARY=$PWD/rasr_amd/librasr_amd_lab.so
for g in 16x8 8x8 32x8 8x4 4x8 16x4 32x4 16x10 125x4 16x40; do
  AMX_TUNING=group=$g python bench.py --workload nn-pipeline --precision f16mx --steps 8 --warmup 2 --no-cpu-baseline --no-configs 2>&1 | grep "^{" | tail -1 | \
   python -c "import sys,json; d=json.loads(sys.stdin.readline()); s=d['stages']; print('group %6s  step %.3f ms  output layer %.3f ms  mean gemm %.3f ms' % ('$g', d['ms_per_step'], s['ffnn_gemm_max']['avg_ms'], s['ffnn_gemm']['avg_ms']))"
done | (mkdir -p gpurun_out/r04; tee gpurun_out/r04/mx_group_sweep.log)
