#!/bin/bash
# tools/pmc_run.sh <tag> <counters...> -- <command...>
# one rocprofv3 --pmc pass (kernel-trace only, as gpurun requires); CSV summary -> gpurun_out/pmc_<tag>.txt
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
out=/tmp/pmc_$tag
rm -rf $out
timeout ${PMC_TIMEOUT:-180} rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d $out -- "$@" > /tmp/pmc_$tag.log 2>&1
tail -5 /tmp/pmc_$tag.log > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log
f=$(find $out -name "*counter_collection.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/pmc_summary.py "$f" "${PMC_FILTER:-}" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt 2>&1
