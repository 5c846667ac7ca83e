cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests/test_gmm_gpu.py -m gpu -x -q -k "best_density_of_the_aligned or u8" 2>&1 | tail -4
for bd in u32 aligned u32 aligned; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-configs --best-density $bd 2>&1 | grep "^{" | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); s=d['stages']; print('$bd', d['value'], d['ms_per_step'], 'gmm', s['gmm']['avg_ms'], 'bestd', s.get('gmm_best_density',{}).get('avg_ms'), 'acc', s['gmm_accumulate']['avg_ms'], 'gemm_max', s['ffnn_gemm_max']['avg_ms'], 'mfcc', s['mfcc']['avg_ms'])"
done | tee gpurun_out/r04/pipeline_aligned_ab.log
