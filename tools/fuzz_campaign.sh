cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05
(for f in "fuzz_gmm.py 600 133" "fuzz_frontends.py 600 131" "fuzz_scorers.py 400 132" "fuzz_tied.py 500 134" "fuzz_ffnn.py 60 136" "fuzz_more.py 150 135" "fuzz_backend.py 150 137"; do echo "== tools/$f"; timeout 1500 python tools/$f 2>&1 | grep -v amdgpu.ids | tail -3; done) > gpurun_out/r05/gpu_fuzz_campaign_end.log 2>&1
cat gpurun_out/r05/gpu_fuzz_campaign_end.log
