"""tools/ksplit_fill_probe.py -- one ring-buffer fill of the decoder (amx_ffnn_score_dev, HIP-graph replay) on BASELINE config 4's network:
default order against amx_ffnn_model.tuning ksplit=4 (split-K across workgroups for passes of at most 256 frames), same process,
alternating, three rounds.  profiles/r05/ksplit_fill.log."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rasr_amd
from tests import synth
ctx = rasr_amd.Context(0); ctx.use_torch_stream()
dims = [440] + [2048] * 6 + [10000]
Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
nns = {t: rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="f16mx", tuning=t) for t in (None, "ksplit=4")}
for T in (64, 256, 512, 1024):
    x = torch.randn((T, 440), device="cuda")
    sc = torch.empty((T, 10000), device="cuda")
    for rnd in range(3):
        for tun, nn in nns.items():
            for _ in range(5): nn.score_dev(x, 440, T, sc)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 300
            for _ in range(n): nn.score_dev(x, 440, T, sc)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            print("T=%d tuning=%s: %.4f ms per fill = %.2f M frames/s" % (T, tun, dt * 1e3, T / dt / 1e6))
