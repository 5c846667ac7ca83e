#!/bin/bash
# tools/tied_ab.sh -- BASELINE config 3, tied instance (4096 shared densities x 10 000 states, batch 256): variants that change the COUNT of
# survivors the pruned scorer walks, not its schedule (round-5 review, item 4).  Per variant: the tied tests against the oracle (bit-exact),
# bench.py --workload gmm-tied in steady state, rocprofv3 kernel averages.  Libraries: tools/build/librasr_amd_near{16,32,128}.so =
# -DAMX_TIED_NEAR=16 | 32 | 128 (near densities per frame behind the bounds U; default 64 since round 6), built before the gpurun call by
#   for n in 16 32 128; do make -C rasr_amd/csrc OBJDIR=build_near$n OUT=../../tools/build/librasr_amd_near$n.so EXTRA=-DAMX_TIED_NEAR=$n; done
# (git-ignored; builds without 64 classes take tied_near_kernel instead of the distance kernel's atomic minima).  Writes gpurun_out/r06/tied_ab.log.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r06/tied_ab.log
mkdir -p $root/gpurun_out/r06
: > $out
variant() {  # label, library ("" = product), gmm tuning ("" = none)
    echo "== $1" >> $out
    lib=$2; tun=$3
    ( cd $root
      if [ -n "$lib" ]; then export AMX_LIBRARY=$root/$lib; fi
      timeout 600 python -m pytest tests/test_gmm_gpu.py tests/test_gmm_contract_gpu.py -x -q -k "tied" 2>&1 | tail -1 >> $out
      python bench.py --workload gmm-tied --steps 200 --warmup 100 --no-cpu-baseline ${tun:+--gmm-tuning $tun} 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
print('   %.4f ms per 256 frames  (%s frames/s)  frac %s  surviving fraction %s  %s' % (d['ms_per_step'], d['value'], r.get('frac'), r.get('surviving_fraction'), r.get('time_is', r.get('kernel'))))" >> $out
      cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tied
      rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tied -- python $root/bench.py --workload gmm-tied --no-cpu-baseline --steps 40 --warmup 5 ${tun:+--gmm-tuning $tun} > /dev/null 2>&1
      f=$(find /tmp/prof_tied -name "*kernel_stats.csv" | head -1)
      [ -n "$f" ] && python3 - "$f" >> $out <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    if "rocclr" in r["Name"] or "at::" in r["Name"]:
        continue
    print("      %-44s calls %4s avg %8.1f us  %5s %%" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
    )
}
variant "default: 64 near densities per frame (round 6), pruned scorer" "" ""
variant "32 near densities (the default until round 6)" tools/build/librasr_amd_near32.so ""
variant "16 near densities (looser bounds, a quarter of the rows)" tools/build/librasr_amd_near16.so ""
variant "128 near densities (tighter bounds, twice the rows)" tools/build/librasr_amd_near128.so ""
variant "tied_prune=0: the dense (min,+) tile kernel" "" "tied_prune=0"
cat $out
