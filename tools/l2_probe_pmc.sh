#!/bin/bash
# tools/l2_probe_pmc.sh <outdir> -- tools/build/l2_probe once plain (rates), once under rocprofv3 --pmc (L2 hit rate of every launch, in
# launch order: the odd launches are the 256-group warm-ups).  Counters in their own pass, kernel-trace only (gpurun's rule).
out=${1:-gpurun_out/r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
$root/tools/build/l2_probe > $root/$out/l2_probe.log 2>&1 || { echo "l2_probe failed"; exit 1; }
rm -rf /tmp/l2pmc
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d /tmp/l2pmc -- $root/tools/build/l2_probe > /tmp/l2pmc.log 2>&1
f=$(find /tmp/l2pmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] || { echo "no counter file"; tail -5 /tmp/l2pmc.log; exit 1; }
python3 - "$f" >> $root/$out/l2_probe.log <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    d.setdefault(int(r['Dispatch_Id']), {'k': r['Kernel_Name'].split('(')[0]})[r['Counter_Name']] = float(r['Counter_Value'])
print("L2 hit rate per launch (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum; launch order = the table above, each preceded by its warm-up):")
for i, (k, v) in enumerate(sorted(d.items())):
    h, m = v.get('TCC_HIT_sum', 0), v.get('TCC_MISS_sum', 0)
    print("  launch %2d %-14s %s  hit %.4g  miss %.4g  req %.4g  hit rate %.3f" % (i, v['k'][:14], "warm-up" if i % 2 == 0 else "timed  ", h, m, v.get('TCC_REQ_sum', 0), h / max(h + m, 1)))
PY
cat $root/$out/l2_probe.log
