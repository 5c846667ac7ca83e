"""tools/output_layer_probe.py [tuning ...] -- time of the OUTPUT layer (2048 -> 10000) of BASELINE config 4's network in f16mx (PREC=...) at
the batch sizes of a decoder's fill and of config 4, per tuning string (tile=N applies to every layer: only the largest layer's
HIP-event time is printed, the pass time beside it)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, ".")
import rasr_amd
from tests import synth
ctx = rasr_amd.Context(0); ctx.use_torch_stream()
prec = os.environ.get("PREC", "f16mx")
dims = [440] + [2048] * 6 + [10000]
Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
for tun in (("default",) if len(sys.argv) < 2 else sys.argv[1:]):
    t = None if tun == "default" else tun
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision=prec, tuning=t)
    for T in (256, 512, 1024, 1536, 2048):
        x = torch.randn((T, 440), device="cuda"); sc = torch.empty((T, 10000), device="cuda")
        def run(n):
            for _ in range(n):
                nn.score_dev(x, 440, T, sc)
        run(20)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(200)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 200
        ctx.profile(True); ctx.profile_reset()
        run(50)
        torch.cuda.synchronize()
        gm = ctx.profile_get("ffnn_gemm_max")
        ctx.profile(False)
        print("%s tuning=%-10s T=%4d: output layer %7.2f us (events)  pass wall %.4f ms" % (prec, tun, T, gm[0] * 1e3, wall * 1e3), flush=True)
