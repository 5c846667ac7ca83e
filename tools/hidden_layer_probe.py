"""tools/hidden_layer_probe.py [tuning ...] -- time of ONE 2048 x 2048 hidden layer of the f16mx GEMM at the batch sizes of BASELINE config 4
(1024 frames) and of a decoder's buffer fill (256), per tuning string.  Network 2048 -> 6 x 2048 -> 64: six identical hidden launches
and a small output layer; hidden = (sum of the per-launch HIP-event times of a pass - the output layer's) / 6.
AMX_LIBRARY=tools/build/librasr_amd_lab.so for the mx_dbg ablations (8 no matrix instructions, 16 no operand DMA after the prologue,
64 every workgroup streams one of 8 tiles: all operands L2 hits)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, ".")
import rasr_amd
from tests import synth
ctx = rasr_amd.Context(0); ctx.use_torch_stream()
prec = os.environ.get("PREC", "f16mx")
H = int(os.environ.get("H", "2048"))   # width of the hidden layers (K = N = H)
dims = [H] * 7 + [64]
Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
for tun in (("default",) if len(sys.argv) < 2 else sys.argv[1:]):
    t = None if tun == "default" else tun
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision=prec, tuning=t)
    for T in (256, 512, 1024, 2048):
        x = torch.randn((T, H), device="cuda"); sc = torch.empty((T, 64), device="cuda")
        def run(n):
            for _ in range(n):
                nn.score_dev(x, H, T, sc)
        run(5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(200)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 200
        ctx.profile(True); ctx.profile_reset()
        run(50)
        torch.cuda.synchronize()
        g = ctx.profile_get("ffnn_gemm"); gm = ctx.profile_get("ffnn_gemm_max")
        ctx.profile(False)
        n_per = g[1] / 50.0
        hid = (g[0] * n_per - gm[0]) / (n_per - 1)
        print("H=%d %s tuning=%-28s T=%4d: hidden layer %7.2f us (events)  pass wall %.4f ms  = %.2f us per layer incl. gaps" % (H, prec, tun, T, hid * 1e3, wall * 1e3, wall * 1e6 / 7), flush=True)
