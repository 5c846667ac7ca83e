#!/bin/bash
# tools/power_probe.sh -- socket power and shader clock (rocm-smi, every 0.7 s) while a workload runs for several seconds: is a kernel
# bound by the 1400 W package power cap (sclk throttled below its 2.4 GHz ceiling) rather than by anything in its instruction stream?
# Run on the GPU box from the repository root; writes gpurun_out/r05b/power_probe.log (copied to profiles/r05/).
mkdir -p gpurun_out/r05b
L=gpurun_out/r05b/power_probe.log
rm -f $L
sample() {  # $1 = label, rest = command
  echo "== $1" >> $L
  shift
  "$@" > gpurun_out/r05b/pp_out.txt 2>&1 &
  local PID=$!
  while kill -0 $PID 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' ' >> $L; echo >> $L; sleep 0.7
  done
  grep -v amdgpu.ids gpurun_out/r05b/pp_out.txt | tail -1 | cut -c1-400 >> $L
}
echo "columns: sclk, socket power (W); cap: $(rocm-smi --showmaxpower 2>&1 | grep -o 'Power (W): [0-9.]*')" >> $L
sample "output layer 2048 -> 10000, 47 952 frames, f16mx, tile=8 (ping-pong, the default): 3000 passes" python tools/out_layer_time.py tile=8 47952 3000
sample "the same, tile=9 (one self-pipelined wave per SIMD)" python tools/out_layer_time.py tile=9 47952 3000
sample "the same, tile=2 (round 4's kernel: eight waves in phase)" python tools/out_layer_time.py tile=2 47952 3000
sample "bench.py --workload nn-pipeline --steps 600 (MFCC -> ctx11 -> FFNN 440-6x2048-10000)" python bench.py --workload nn-pipeline --no-cpu-baseline --no-configs --steps 600 --warmup 3
sample "bench.py --workload gmm --steps 1500 (fused GMM scorer, cfg3-cart)" python bench.py --workload gmm --no-cpu-baseline --no-configs --steps 1500 --warmup 3
sample "bench.py --workload pipeline --steps 400 (headline: both scorers)" python bench.py --no-cpu-baseline --no-configs --steps 400 --warmup 3
sample "bench.py --workload nn-pipeline --precision bf16 --steps 600" python bench.py --workload nn-pipeline --precision bf16 --no-cpu-baseline --no-configs --steps 600 --warmup 3
cat $L
