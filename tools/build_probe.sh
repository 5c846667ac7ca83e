#!/bin/bash
# builds tools/build/gemm_probe (see tools/gemm_probe.hip)
set -e
cd "$(dirname "$0")/.."
make -s -C rasr_amd/csrc
mkdir -p tools/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I include -I rasr_amd/csrc -c tools/gemm_probe.hip -o tools/build/gemm_probe.o
hipcc --offload-arch=gfx950 tools/build/gemm_probe.o rasr_amd/csrc/build/api.o rasr_amd/csrc/build/stats.o -o tools/build/gemm_probe
# tools/build/ceilings: measured HBM / MFMA / VALU ceilings of the box (tools/ceilings.hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/ceilings.hip -o tools/build/ceilings
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/write_probe.hip -o tools/build/write_probe
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/valu_rates.hip -o tools/build/valu_rates
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/gather_probe.hip -o tools/build/gather_probe
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/feed_probe.hip -o tools/build/feed_probe
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/l2_probe.hip -o tools/build/l2_probe
# the lab build of the library (time stamps inside the kernels; tools/mx_timeline.py, tools/fused_timeline.py): rebuilt with the probes so
# that it can never lag behind the ABI the tools bind (round 5 committed two tracebacks: the lab library predated a symbol)
make -s -j4 -C rasr_amd/csrc OBJDIR=build_lab OUT=../../tools/build/librasr_amd_lab.so EXTRA=-DAMX_LAB CHECK=-
rm -f tools/build/*.bc tools/build/*.hipi tools/build/*.out tools/build/*.s tools/build/*.resolution.txt tools/build/*gfx950.o tools/build/*x86_64*
