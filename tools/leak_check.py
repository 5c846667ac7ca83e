"""tools/leak_check.py -- create / use / destroy every handle type repeatedly and watch the free device memory (hipMemGetInfo through torch):
a leak of device buffers, graphs or workspaces shows up as a steady decline."""
import gc
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rasr_amd  # noqa: E402
from tests import synth  # noqa: E402

ctx = rasr_amd.Context(0)
ctx.use_torch_stream()


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


model = synth.gmm_cart(2000, 16, 16, 40, seed=1, pooled=True)
tied = synth.gmm_tied(500, 256, 40, seed=2, pooled=True)
Ws, bs, acts, logp = synth.ffnn([440, 1024, 1024, 3000], seed=3)
x = torch.randn((3000, 40), device="cuda")
xin = torch.randn((3000, 440), device="cuda")
pcm = synth.waveform(48000, seed=4)
base = None
for rep in range(12):
    for kind in ("diagonal-maximum", "SIMD-diagonal-maximum", "batch-diagonal-maximum-int", "diagonal-sum", "preselection-batch-float",
                 "preselection-batch-int", "batch-diagonal-maximum-float"):
        for m in (model, tied):
            if kind.startswith(("preselection", "batch-diagonal-maximum-float")) and m is tied:
                continue
            sc = rasr_amd.GmmFeatureScorer(ctx, m, feature_scorer_type=kind)
            if kind.startswith("preselection"):
                sc.set_preselection(64, 8, 3, 40000.0)
            nm = len(m["mix_offsets"]) - 1
            s = torch.empty((3000, nm), device="cuda")
            for T in (256, 256, 256, 3000, 256):          # small passes (graph replay), a large one (workspace growth), small again
                sc.score_dev(x, T, s, None)
            del sc, s
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="bf16")
    s = torch.empty((3000, 3000), device="cuda")
    for T in (1024, 1024, 1024, 3000, 1024):
        nn.score_dev(xin, 440, T, s)
    out = torch.empty((3000, 3000), device="cuda")
    nn.forward_dev(xin, 440, 3000, out, top="softmax")
    nn.forward_dev(xin, 440, 1024, out, top="linear")
    del nn, s, out
    for fe_kw in (dict(), dict(front_end="mfplp", nr_autocorrelation_coefficients=13, nr_cepstrum_coefficients=13, normalize=True)):
        fe = rasr_amd.MfccExtractor(ctx, **fe_kw)
        fe.run(pcm)
        del fe
    fe = rasr_amd.MfccExtractor.plp(ctx)
    fe.run(pcm)
    del fe
    gt = rasr_amd.GammatoneExtractor(ctx, channels=68, max_freq=7500.0, si_length=9, si_shift=4, power=0.1, n_ceps=12)
    gt.run(pcm)
    del gt
    gc.collect()
    torch.cuda.empty_cache()
    f = free_mb()
    if rep == 1:
        base = f
    print("round %2d free %.1f MiB" % (rep, f))
drift = base - free_mb()
print("drift after warm-up: %.1f MiB" % drift)
sys.exit(1 if drift > 64 else 0)
