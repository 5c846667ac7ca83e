cd /root/repo; mkdir -p gpurun_out/r04
(python tools/mfcc_timeline.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r04/mfcc_timeline.log
(python tools/mfcc_timeline.py fft=r16 2>&1 | grep -v amdgpu.ids) > gpurun_out/r04/mfcc_r16_timeline.log
cat gpurun_out/r04/mfcc_r16_timeline.log | tail -3
