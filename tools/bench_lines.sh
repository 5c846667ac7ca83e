#!/bin/bash
# tools/bench_lines.sh <round> -- only the bench.py JSON lines of tools/profile_all.sh (no profiler runs): gpurun_out/<round>/<workload>_bench.log
round=${1:-r05}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$round
mkdir -p $out
line() { name=$1; shift; python $root/bench.py "$@" 2>/dev/null | tail -1 > $out/${name}_bench.log; }
python $root/bench.py --no-cpu-baseline --no-configs > /dev/null 2>&1
line pipeline --steps 10 --warmup 2
line pipeline-streamed --ingest streamed --steps 20 --warmup 2 --no-cpu-baseline --no-configs
line pipeline-bf16 --precision bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-configs
line pipeline-bf16x3 --precision bf16x3 --steps 10 --warmup 2 --no-cpu-baseline --no-configs
line nn-pipeline --workload nn-pipeline --steps 10 --warmup 2 --no-cpu-baseline
line nn-pipeline-bf16x3 --workload nn-pipeline --precision bf16x3 --steps 10 --warmup 2 --no-cpu-baseline
line nn-pipeline-bf16 --workload nn-pipeline --precision bf16 --steps 10 --warmup 2 --no-cpu-baseline
line nn-pipeline-fp32 --workload nn-pipeline --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline
line gmm-train --workload gmm-train --steps 5 --warmup 2 --no-cpu-baseline
line gmm-trained --workload gmm-trained --steps 5 --warmup 2 --no-cpu-baseline
line mfcc --workload mfcc --steps 20 --warmup 2 --no-cpu-baseline
line gmm --workload gmm --steps 50 --warmup 5 --no-cpu-baseline
line gmm-tied --workload gmm-tied --steps 20 --warmup 3
line nn --workload nn --steps 50 --warmup 5 --no-cpu-baseline
line nn-bf16x3 --workload nn --precision bf16x3 --steps 50 --warmup 5 --no-cpu-baseline
line nn-bf16 --workload nn --precision bf16 --steps 50 --warmup 5 --no-cpu-baseline
line mfcc-plp --workload mfcc --front-end plp --steps 8 --warmup 2 --no-cpu-baseline
line mfcc-mfplp --workload mfcc --front-end mfplp --steps 8 --warmup 2 --no-cpu-baseline
line mfcc-gammatone --workload mfcc --front-end gammatone --steps 3 --warmup 1 --no-cpu-baseline
AMX_BENCH_FORCE_DIST=1 python $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | grep "^{\"metric\"" | tail -1 > $out/force_dist_bench.log
ls $out
