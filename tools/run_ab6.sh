cd /root/repo
python -m pytest tests/test_mfcc_gpu.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
for lib in librasr_amd_old.so librasr_amd.so; do
AMX_LIBRARY=$PWD/rasr_amd/$lib python bench.py --workload mfcc --steps 20 --warmup 3 --no-cpu-baseline --no-configs 2>&1 | grep "^{" | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['ms_per_step'])"
done; done
