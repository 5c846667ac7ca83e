cd /root/repo; mkdir -p gpurun_out/r04
python -m pytest tests/test_mfcc_gpu.py -m gpu -x -q 2>&1 | tail -3
for t in "prefetch=1" "prefetch=1" "prefetch=1"; do
python bench.py --workload mfcc --steps 20 --warmup 3 --no-cpu-baseline --no-configs ${t:+--mfcc-tuning $t} 2>&1 | grep "^{" | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('tuning [$t]', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done | tee gpurun_out/r04/mfcc_ab5.log
python tools/mfcc_timeline.py 2>&1 | grep -v amdgpu.ids
