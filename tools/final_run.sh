root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r06
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests -m gpu -q > $out/gpu_tests_lease5.log 2>&1; echo "suite rc=$?" >> $out/gpu_tests_lease5.log
tail -3 $out/gpu_tests_lease5.log
ONLY='^gmm-tied$' bash tools/profile_all.sh r06 > /dev/null 2>&1
cat $out/gmm-tied_bench.log | cut -c1-300
python bench.py 2>/dev/null | grep '^{"metric"' | tail -1 > $out/default_bench.log
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06/default_bench.log').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['epoch_reduce'])
print([(k, v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in (d.get('configs') or {}).items()][:12])
PY
bash tools/fuzz_campaign.sh r06 8100 > /dev/null 2>&1; tail -30 $out/gpu_fuzz_campaign.log
