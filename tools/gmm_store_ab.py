"""A/B: how much of gmm_fused_kernel's time are its stores?  The same 63 936 x 10 000 x 16 pass with the best-density matrix (u32, 2.56 GB)
as bytes (amx_gmm_score_stats_u8_dev, 0.64 GB) and without it (scores only).  python tools/gmm_store_ab.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rasr_amd
from tests import synth

ctx = rasr_amd.Context(0)
ctx.use_torch_stream()
model = synth.gmm_cart(10000, 16, 16, 40, seed=5, pooled=True)
gmm = rasr_amd.GmmFeatureScorer(ctx, model)
T, M = 63936, 10000
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((T, 40), device="cuda", generator=g) * 3.0
scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
bestd = torch.empty((T, M), dtype=torch.int32, device="cuda")
state = torch.empty((T,), dtype=torch.int32, device="cuda")
counts = torch.zeros((M,), dtype=torch.int64, device="cuda")
ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
bestd8 = torch.empty((T, M), dtype=torch.uint8, device="cuda")
for name, bd in (("scores + best density u32", bestd), ("scores + best density u8", bestd8), ("scores only", None)) * 2:
    for _ in range(2):
        gmm.score_stats_dev(x, T, scores, bd, state, counts, ssum)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        gmm.score_stats_dev(x, T, scores, bd, state, counts, ssum)
    e1.record()
    torch.cuda.synchronize()
    print("%-28s %.3f ms per pass" % (name, e0.elapsed_time(e1) / 6))
