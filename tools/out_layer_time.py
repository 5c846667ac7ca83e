"""tools/out_layer_time.py [tuning] -- time of the config-5 output layer alone (2048 -> 10000, 47 952 frames, f16mx), HIP events around
20 launches after 5 warm-up launches; AMX_LIBRARY selects the build (A/B runs of K-loop variants)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rasr_amd  # noqa: E402
from tests import synth  # noqa: E402

tuning = sys.argv[1] if len(sys.argv) > 1 else None
T = int(sys.argv[2]) if len(sys.argv) > 2 else 47952
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = rasr_amd.Context(0)
ctx.use_torch_stream()
Ws, bs, acts, logp = synth.ffnn([2048, 10000], seed=7)
x = torch.from_numpy(np.random.Generator(np.random.PCG64(1)).standard_normal((T, 2048)).astype(np.float32)).cuda()
nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="f16mx", tuning=tuning)
s = torch.empty((T, 10000), dtype=torch.float32, device="cuda")
for _ in range(5):
    nn.score_dev(x, 2048, T, s)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
clk = torch.zeros((2, 2), dtype=torch.int64, device="cuda")
ctx.device_clocks(clk[0])
ev[0].record()
for _ in range(N):
    nn.score_dev(x, 2048, T, s)
    if N > 100:
        torch.cuda.synchronize()
ev[1].record()
ctx.device_clocks(clk[1])
torch.cuda.synchronize()
ck = clk.cpu().numpy().astype(np.float64)
ghz = (ck[1, 0] - ck[0, 0]) / max(ck[1, 1] - ck[0, 1], 1.0) * 0.1
ctx.profile(True)
ctx.profile_reset()
for _ in range(10):
    nn.score_dev(x, 2048, T, s)
torch.cuda.synchronize()
ms_g, n_g = ctx.profile_get("ffnn_gemm")
print("%-28s %-12s T=%d  %.4f ms per pass, gemm_mx_kernel %.4f ms per launch (HIP events, %d launches)  checksum %.6e  shader clock %.3f GHz" % (os.path.basename(os.environ.get("AMX_LIBRARY", "librasr_amd.so")), tuning, T,
                                                                    ev[0].elapsed_time(ev[1]) / N, ms_g, n_g, float(s[:64].double().sum()), ghz))
