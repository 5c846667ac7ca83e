run() { python bench.py "$@" --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['ms_per_step'], {k:v.get('avg_ms') for k,v in d.get('stages',{}).items() if isinstance(v,dict)})"; }
for fe in mfcc mfplp plp; do
  for lib in tree alt tree alt; do
    if [ $lib = alt ]; then export AMX_LIBRARY=$GRAFT_REPO_ROOT/tools/build/librasr_amd_old.so; else unset AMX_LIBRARY; fi
    echo -n "$fe $lib"; run --workload mfcc --front-end $fe
  done
done
unset AMX_LIBRARY
timeout 900 python -m pytest tests/test_mfcc_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -3
