"""tools/graph_probe.py -- HIP-graph replay (tuning graph=1, the default) against plain launches (graph=0) for the small-batch passes, in the
two ways a caller can drive them: passes enqueued back to back (one synchronisation at the end: bench.py's loop, a training pass) and one
synchronisation per pass (a decoder that needs the scores before it goes on)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, ".")
import rasr_amd
from tests import synth
ctx = rasr_amd.Context(0); ctx.use_torch_stream()

def timeit(fn, n):
    fn(); fn(); fn(); fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); a = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        fn(); torch.cuda.synchronize()
    b = (time.perf_counter() - t0) / n
    return a * 1e3, b * 1e3

dims = [440] + [2048] * 6 + [10000]
Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
for prec in ("f16mx", "bf16"):
    for g in ("graph=1", "graph=0"):
        nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision=prec, tuning=g)
        for T in (256, 1024):
            x = torch.randn((T, 440), device="cuda"); sc = torch.empty((T, 10000), device="cuda")
            a, b = timeit(lambda: nn.score_dev(x, 440, T, sc), 300)
            print("nn %-5s %s T=%4d: back to back %.4f ms, synchronised per pass %.4f ms" % (prec, g, T, a, b), flush=True)
        del nn
model = synth.gmm_tied(10000, 4096, 40, seed=5, pooled=True)
x = torch.randn((256, 40), device="cuda"); sc = torch.empty((256, 10000), device="cuda"); best = torch.empty((256, 10000), dtype=torch.int32, device="cuda")
for g in ("graph=1", "graph=0"):
    s = rasr_amd.GmmFeatureScorer(ctx, model, tuning=g)
    a, b = timeit(lambda: s.score_dev(x, 256, sc, best), 300)
    print("gmm-tied %s T= 256: back to back %.4f ms, synchronised per pass %.4f ms" % (g, a, b), flush=True)
    del s
model = synth.gmm_cart(10000, 16, 16, 40, seed=6, pooled=True)
for g in ("fused_pack=0,graph=1", "fused_pack=0,graph=0", "graph=0"):
    s = rasr_amd.GmmFeatureScorer(ctx, model, tuning=g)
    a, b = timeit(lambda: s.score_dev(x, 256, sc, best), 300)
    print("gmm-cart %s T= 256: back to back %.4f ms, synchronised per pass %.4f ms" % (g, a, b), flush=True)
    del s
