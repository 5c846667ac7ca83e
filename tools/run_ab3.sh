cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests/test_gmm_gpu.py -m gpu -x -q -k "u8" 2>&1 | tail -2
python tools/gmm_store_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/gmm_store_ab.log
for bd in u8 u32 u8 u32; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-configs --best-density $bd 2>&1 | grep "^{" | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); s=d['stages']; print('$bd', d['value'], d['ms_per_step'], 'gmm', s['gmm']['avg_ms'], 'acc', s['gmm_accumulate']['avg_ms'], 'gemm_max', s['ffnn_gemm_max']['avg_ms'])"
done | tee gpurun_out/r04/pipeline_u8_ab.log
