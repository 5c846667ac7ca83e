#!/bin/bash
# tools/profile_all.sh <round> -- regenerates the artefacts under profiles/<round> on a GPU box:
#   <workload>_bench.log          the bench.py JSON line (pipeline = the default run, with cpu_baseline and the secondary configs)
#   <workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the same command
#   pmc/*.txt                     separate --pmc passes (kernel-trace only): FETCH_SIZE / WRITE_SIZE, SQ counters, MFMA counters
#   gpu_tests.log / gpu_fuzz.log  the -m gpu suite and one campaign of every GPU fuzzer (tools/fuzz_*.py)
#   force_dist_bench.log          the default workload with the RCCL path forced on (world size 1)
#   gemm_comparator.json          torch (hipBLASLt / rocBLAS) bf16 GEMM of the output-layer shape on the same box: measurement only
#   gather_probe.log              scattered 256-byte rows out of L2 by request shape; cost of a scattered 64-lane gather (tools/gather_probe.hip)
#   gemm_probe.log                ablation table of the pipelined GEMM (tools/gemm_probe.hip: full / no MFMA / no DMA / L2-resident operands)
# usage (through gpurun):  tools/profile_all.sh r04 ; results land in gpurun_out/<round>/ -> copy to profiles/<round>/
round=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$round
mkdir -p $out/pmc
cd /tmp && export TMPDIR=/tmp
# (AMX_BENCH_NO_SMI=1 under the profiler: bench.py's 2.5 s rocm-smi stretch behind the timed region would only add launches to the traces)
# every tool's exit status is kept: a tool that exits non-zero (or leaves a Python traceback in its log) is listed in $out/FAILED and
# the script exits 1 -- a traceback must never be committed as a profile again (round 5: mx_timeline.log, fused_timeline.log)
: > $out/FAILED
must() {  # logfile, command...
    log=$1; shift
    "$@" > $log 2>&1
    rc=$?
    if [ $rc -ne 0 ] || grep -q "^Traceback (most recent call last)" $log; then
        echo "$log: rc=$rc: $*" >> $out/FAILED
        mv $log $log.FAILED
    fi
}
only() { [ -z "$ONLY" ] || [[ "$1" =~ $ONLY ]]; }  # ONLY=<regex>: just the workloads whose name matches (and none of the suites)
run_stats() {  # name, bench args...
    only $1 || return 0
    name=$1; shift
    python $root/bench.py "$@" 2>/tmp/bench_$name.err | tail -1 > $out/${name}_bench.log
    grep -q '^{"metric"' $out/${name}_bench.log || { echo "$out/${name}_bench.log: no JSON line (bench.py $*): $(tail -1 /tmp/bench_$name.err)" >> $out/FAILED; mv $out/${name}_bench.log $out/${name}_bench.log.FAILED; }
    rm -rf /tmp/prof_$name
    AMX_BENCH_NO_SMI=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $root/bench.py "$@" --no-cpu-baseline --no-configs > /tmp/prof_$name.log 2>&1
    f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then cp $f $out/${name}_kernel_stats.csv; else echo "$name: rocprofv3 left no kernel_stats.csv" >> $out/FAILED; fi
}
run_pmc() {  # name, suffix, counters..., -- bench args...
    only $1 || return 0
    name=$1; suf=$2; shift 2
    ctrs=()
    while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
    shift
    rm -rf /tmp/pmc_${name}_$suf
    AMX_BENCH_NO_SMI=1 timeout 300 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d /tmp/pmc_${name}_$suf -- python $root/bench.py "$@" --no-cpu-baseline --no-configs > /tmp/pmc_${name}_$suf.log 2>&1
    f=$(find /tmp/pmc_${name}_$suf -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $root/tools/pmc_summary.py $f > $out/pmc/${name}_$suf.txt
}
python $root/bench.py --no-cpu-baseline --no-configs > /dev/null 2>&1   # warm the box / caches
run_stats pipeline --steps 10 --warmup 2
run_stats pipeline-streamed --ingest streamed --steps 20 --warmup 2 --no-cpu-baseline --no-configs
run_stats pipeline-bf16 --precision bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-configs
run_stats pipeline-bf16x3 --precision bf16x3 --steps 10 --warmup 2 --no-cpu-baseline --no-configs
run_stats nn-pipeline --workload nn-pipeline --steps 10 --warmup 2 --no-cpu-baseline
run_stats nn-pipeline-bf16x3 --workload nn-pipeline --precision bf16x3 --steps 10 --warmup 2 --no-cpu-baseline
run_stats nn-pipeline-bf16 --workload nn-pipeline --precision bf16 --steps 10 --warmup 2 --no-cpu-baseline
run_stats nn-pipeline-fp32 --workload nn-pipeline --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline
run_stats gmm-train --workload gmm-train --steps 5 --warmup 2 --no-cpu-baseline
run_stats gmm-trained --workload gmm-trained --steps 5 --warmup 2 --no-cpu-baseline
run_stats mfcc --workload mfcc --steps 20 --warmup 2 --no-cpu-baseline
run_stats gmm --workload gmm --steps 400 --warmup 200 --no-cpu-baseline
run_stats gmm-tied --workload gmm-tied --steps 200 --warmup 100
run_stats nn --workload nn --steps 400 --warmup 200 --no-cpu-baseline
run_stats nn-bf16x3 --workload nn --precision bf16x3 --steps 400 --warmup 200 --no-cpu-baseline
run_stats nn-bf16 --workload nn --precision bf16 --steps 400 --warmup 200 --no-cpu-baseline
run_stats mfcc-plp --workload mfcc --front-end plp --steps 8 --warmup 2 --no-cpu-baseline
run_stats mfcc-mfplp --workload mfcc --front-end mfplp --steps 8 --warmup 2 --no-cpu-baseline
run_stats mfcc-gammatone --workload mfcc --front-end gammatone --steps 3 --warmup 1 --no-cpu-baseline
run_pmc pipeline fetch FETCH_SIZE -- --steps 3 --warmup 1
run_pmc pipeline write WRITE_SIZE -- --steps 3 --warmup 1
run_pmc pipeline-bf16 fetch FETCH_SIZE -- --precision bf16 --steps 3 --warmup 1
run_pmc pipeline-bf16 write WRITE_SIZE -- --precision bf16 --steps 3 --warmup 1
run_pmc mfcc fetch FETCH_SIZE -- --workload mfcc --steps 3 --warmup 1
run_pmc mfcc write WRITE_SIZE -- --workload mfcc --steps 3 --warmup 1
run_pmc mfcc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS -- --workload mfcc --steps 3 --warmup 1
run_pmc mfcc sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY -- --workload mfcc --steps 3 --warmup 1
run_pmc nn-pipeline l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum -- --workload nn-pipeline --steps 3 --warmup 1
run_pmc nn-pipeline fetch FETCH_SIZE -- --workload nn-pipeline --steps 3 --warmup 1
run_pmc nn-pipeline write WRITE_SIZE -- --workload nn-pipeline --steps 3 --warmup 1
run_pmc nn-pipeline mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA -- --workload nn-pipeline --steps 3 --warmup 1
run_pmc nn-pipeline sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY -- --workload nn-pipeline --steps 3 --warmup 1
run_pmc nn-pipeline-bf16 mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA -- --workload nn-pipeline --precision bf16 --steps 3 --warmup 1
run_pmc gmm-train sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -- --workload gmm-train --steps 3 --warmup 1
run_pmc gmm-train sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -- --workload gmm-train --steps 3 --warmup 1
run_pmc gmm-tied sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS -- --workload gmm-tied --steps 5 --warmup 2
run_pmc gmm-tied sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY -- --workload gmm-tied --steps 5 --warmup 2
run_pmc gmm-tied fetch FETCH_SIZE -- --workload gmm-tied --steps 5 --warmup 2
run_pmc gmm-tied write WRITE_SIZE -- --workload gmm-tied --steps 5 --warmup 2
[ -x $root/tools/build/gather_probe ] && $root/tools/build/gather_probe > $out/gather_probe.log 2>&1
if [ -n "$ONLY" ]; then ls -la $out $out/pmc; exit 0; fi
(cd $root && timeout 1500 python -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1)
(cd $root && for f in "fuzz_frontends.py 300 31" "fuzz_scorers.py 300 32" "fuzz_gmm.py 200 33" "fuzz_tied.py 300 34" "fuzz_more.py 100 35" "fuzz_ffnn.py 40 36" "fuzz_backend.py 100 37"; do echo "== tools/$f"; timeout 900 python tools/$f 2>&1 | grep -v amdgpu.ids | tail -2; done > $out/gpu_fuzz.log 2>&1)
AMX_BENCH_FORCE_DIST=1 python $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | grep "^{\"metric\"" | tail -1 > $out/force_dist_bench.log
python $root/tools/gemm_comparator.py > $out/gemm_comparator.json 2>/dev/null
[ -x $root/tools/build/feed_probe ] && $root/tools/build/feed_probe > $out/feed_probe.log 2>&1
if [ -f $root/tools/build/librasr_amd_lab.so ]; then
  (cd $root && must $out/mx_timeline.log python tools/mx_timeline.py 0; must $out/mx_timeline_small.log python tools/mx_timeline.py small; must $out/fused_timeline.log python tools/fused_timeline.py)
else
  echo "tools/build/librasr_amd_lab.so missing (tools/build_probe.sh builds it)" >> $out/FAILED
fi
[ -x $root/tools/build/l2_probe ] && must $out/l2_probe.log $root/tools/build/l2_probe
[ -x $root/tools/build/gemm_probe ] && PROBE_RELU=1 $root/tools/build/gemm_probe xp0 xp8 xp16 xp24 xp64 xp72 xp4 p0 p8 p16 p64 p72 p4 x0 xa0 xp0:16x8 p0:16x8 > $out/gemm_probe.log 2>&1
[ -x $root/tools/build/valu_rates ] && $root/tools/build/valu_rates > $out/valu_rates.log 2>&1
[ -x $root/tools/build/ceilings ] && $root/tools/build/ceilings > $out/ceilings.json 2>/dev/null
python $root/tools/traffic_json.py $out > $out/traffic.json 2>/dev/null
ls -la $out $out/pmc
if [ -s $out/FAILED ]; then echo "profile_all: FAILED tools:"; cat $out/FAILED; exit 1; fi
rm -f $out/FAILED
