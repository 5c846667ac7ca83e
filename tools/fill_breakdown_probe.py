import os, sys, time, numpy as np, torch
sys.path.insert(0, ".")
import rasr_amd
from tests import synth
ctx = rasr_amd.Context(0); ctx.use_torch_stream()
dims = [440] + [2048] * 6 + [10000]
Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
for tun in ((None, "ksplit=4") if len(sys.argv) < 2 else sys.argv[1:]):
    tun = None if tun == "default" else tun
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="f16mx", tuning=tun)
    for T in (64, 256, 512, 1024):
        x = torch.randn((T, 440), device="cuda"); sc = torch.empty((T, 10000), device="cuda")
        for _ in range(5): nn.score_dev(x, 440, T, sc)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): nn.score_dev(x, 440, T, sc)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 200
        ctx.profile(True); ctx.profile_reset()
        for _ in range(50): nn.score_dev(x, 440, T, sc)
        torch.cuda.synchronize()
        g = ctx.profile_get("ffnn_gemm"); gm = ctx.profile_get("ffnn_gemm_max"); pk = ctx.profile_get("ffnn_pack")
        ctx.profile(False)
        n_per = g[1] / 50.0
        print("tuning=%s T=%d: wall %.4f ms | per-launch events: gemm avg %.4f ms x %.1f per pass = %.4f ms (output layer %.4f), pack %.4f" % (tun, T, wall * 1e3, g[0], n_per, g[0] * n_per, gm[0], pk[0]))
