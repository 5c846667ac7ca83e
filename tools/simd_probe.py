"""tools/simd_probe.py -- times the SIMD-diagonal-maximum scorer (config-3 CART model) with and without best-density output.
usage (GPU box): python tools/simd_probe.py [frames]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rasr_amd
from tests import synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = rasr_amd.Context(0)
ctx.use_torch_stream()
model = synth.gmm_cart(10000, 16, 16, 40, seed=5, pooled=True)
sc = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="SIMD-diagonal-maximum")
x = torch.from_numpy(np.random.default_rng(0).standard_normal((T, 40)).astype(np.float32)).cuda()
scores = torch.empty((T, 10000), dtype=torch.float32, device="cuda")
best = torch.empty((T, 10000), dtype=torch.int32, device="cuda")
for name, b in (("scores+best", best), ("scores only", None)):
    for _ in range(3):
        sc.score_dev(x, T, scores, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        sc.score_dev(x, T, scores, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = T * 10000 * (8 if b is not None else 4)
    print("%s: %.3f ms  %.1f M frames/s  %.0f GB/s out" % (name, ms, T / ms / 1e3, by / ms / 1e6))
