#!/bin/bash
# tools/fuzz_campaign.sh [round] [seed base] -- the closing campaign of a round: every GPU fuzzer with fresh seeds, contract mode drawn per case
# where the entry point has one (gmm, tied, scorers, backend, frontends).  Writes gpurun_out/<round>/gpu_fuzz_campaign.log; exits 1 on a mismatch.
cd "$(dirname "$0")/.."; round=${1:-r06}; base=${2:-600}; mkdir -p gpurun_out/$round
out=gpurun_out/$round/gpu_fuzz_campaign.log
(for f in "fuzz_gmm.py 600 $((base+33))" "fuzz_frontends.py 600 $((base+31))" "fuzz_scorers.py 400 $((base+32))" "fuzz_tied.py 500 $((base+34))" "fuzz_ffnn.py 60 $((base+36))" "fuzz_more.py 150 $((base+35))" "fuzz_backend.py 300 $((base+37))"; do echo "== tools/$f"; timeout 1500 python tools/$f 2>&1 | grep -v "amdgpu.ids\|^rasr_amd: amx_ffnn_create" | tail -3; done) > $out 2>&1
cat $out
grep -q "MISMATCH\|Traceback" $out && exit 1
grep -c " 0 mismatches" $out
