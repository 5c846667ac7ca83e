#!/bin/bash
# tools/profile_all.sh -- regenerates the artefacts under profiles/rNN on a GPU box:
#   <workload>_bench.log          the bench.py JSON line (default workload: with cpu_baseline)
#   <workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the same command
#   ceilings.json                 measured HBM / MFMA / VALU ceilings of the box (tools/ceilings.hip)
#   pmc/<workload>_{fetch,write}.txt   FETCH_SIZE / WRITE_SIZE per kernel, separate --pmc passes (kernel-trace only)
# usage (through gpurun):  tools/profile_all.sh r01 ; results land in gpurun_out/<round>/ -> copy to profiles/<round>/
round=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$round
mkdir -p $out/pmc
cd /tmp && export TMPDIR=/tmp
run_stats() {  # name, bench args...
    name=$1; shift
    python $root/bench.py "$@" --no-cpu-baseline > /dev/null 2>&1   # warm the box / caches
    python $root/bench.py "$@" 2>/dev/null | tail -1 > $out/${name}_bench.log
    rm -rf /tmp/prof_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $root/bench.py "$@" --no-cpu-baseline > /tmp/prof_$name.log 2>&1
    f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $out/${name}_kernel_stats.csv
}
run_pmc() {  # name, counter, suffix, bench args...
    name=$1; ctr=$2; suf=$3; shift 3
    rm -rf /tmp/pmc_${name}_$suf
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${name}_$suf -- python $root/bench.py "$@" --no-cpu-baseline > /tmp/pmc_${name}_$suf.log 2>&1
    f=$(find /tmp/pmc_${name}_$suf -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $root/tools/pmc_summary.py $f > $out/pmc/${name}_$suf.txt
}
[ -x $root/tools/build/ceilings ] && $root/tools/build/ceilings > $out/ceilings.json   # tools/build_probe.sh builds it
run_stats pipeline --steps 8 --warmup 2
run_stats nn-pipeline --workload nn-pipeline --steps 8 --warmup 2 --no-cpu-baseline
run_stats mfcc --workload mfcc --steps 8 --warmup 2 --no-cpu-baseline
run_stats mfplp --workload mfcc --front-end mfplp --steps 8 --warmup 2 --no-cpu-baseline
run_stats gmm --workload gmm --steps 50 --warmup 5 --no-cpu-baseline
run_stats gmm-simd --workload gmm --gmm-type SIMD-diagonal-maximum --gmm-frames 65536 --steps 20 --warmup 3 --no-cpu-baseline
run_stats gmm-tied --workload gmm-tied --steps 20 --warmup 3 --no-cpu-baseline
run_stats gmm-train --workload gmm-train --steps 5 --warmup 2 --no-cpu-baseline
run_stats nn --workload nn --steps 50 --warmup 5 --no-cpu-baseline
run_pmc pipeline FETCH_SIZE fetch --steps 3 --warmup 1
run_pmc pipeline WRITE_SIZE write --steps 3 --warmup 1
run_pmc mfcc FETCH_SIZE fetch --workload mfcc --steps 3 --warmup 1
run_pmc mfcc WRITE_SIZE write --workload mfcc --steps 3 --warmup 1
ls -la $out $out/pmc
