"""tools/fuzz_gmm.py [n_cases] [seed] [contract: off | fma | both (default)] -- random diagonal-GMM models / feature batches through every maximum-approximation path of the
library against the oracle, bit for bit: private and pooled covariances, 1..16 and longer mixtures, tied lists, supported and
unsupported dimensions, features with outliers that overflow the f16 screen operand (all-slot frames), duplicated densities.
Every case draws one of the reference's two arithmetics (amx_gmm_model.tuning contract=off | fma) and is held to the oracle library of that
build.  Not part of the test suite (minutes of oracle time); run on a GPU box after touching gmm.hip / gmm_simd.hip."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rasr_amd  # noqa: E402
from oracle import OracleGmm  # noqa: E402
from tests import synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
contracts = {"off": ["off"], "fma": ["fma"]}.get(sys.argv[3] if len(sys.argv) > 3 else "both", ["off", "fma"])
ctx = rasr_amd.Context(0)
bad = 0
for case in range(n_cases):
    dim = int(rng.choice([16, 24, 32, 33, 39, 40, 45, 48, 64, 7, 50]))
    pooled = bool(rng.integers(0, 2))
    kind = rng.choice(["cart", "cart", "cart", "long", "tied"])
    seed = int(rng.integers(1, 1 << 30))
    if kind == "cart":
        model = synth.gmm_cart(int(rng.integers(1, 300)), 1, int(rng.integers(1, 17)), dim, seed=seed, pooled=pooled)
    elif kind == "long":
        model = synth.gmm_cart(int(rng.integers(1, 40)), 17, 60, dim, seed=seed, pooled=pooled)
    else:
        nd = int(rng.integers(8, 200))
        model = synth.gmm_tied(int(rng.integers(4, 120)), nd, dim, seed=seed, pooled=pooled, alpha=float(rng.choice([0.1, 1.0])),
                               k_per_mix=None if rng.integers(0, 2) else int(rng.integers(4, nd + 1)))
    if rng.integers(0, 3) == 0 and model["means"].shape[0] > 2:       # duplicated densities
        model["means"][1] = model["means"][0]
    twist = int(rng.integers(0, 8))
    if twist == 0:      # zero-weight densities: the text reader stores Core::Type<f64>::min for them
        lw = model["log_weight"].copy()
        lw[rng.random(len(lw)) < 0.2] = -1.7976931348623157e+308
        model["log_weight"] = lw
    elif twist == 1:    # variances over eight orders of magnitude
        model["variances"] = (model["variances"] * np.float32(10.0) ** rng.integers(-4, 5, model["variances"].shape)).astype(np.float32)
    elif twist == 2:    # means far from the features
        model["means"] = (model["means"] * np.float32(100.0)).astype(np.float32)
    elif twist == 3:    # nearly equal weights and means: many near ties
        model["means"] = (model["means"][:1] + np.float32(1e-3) * model["means"]).astype(np.float32)
    T = int(rng.choice([1, 3, 63, 64, 65, 255, 256, 257, 700]))
    x = rng.standard_normal((T, dim)).astype(np.float32) * np.float32(rng.choice([0.3, 1.0, 3.0]))
    if rng.integers(0, 3) == 0:
        x[rng.integers(0, T)] *= np.float32(1e4)                      # does not fit the f16 operand: frame keeps all slots
    if rng.integers(0, 6) == 0:
        x[rng.integers(0, T), rng.integers(0, dim)] = np.float32(rng.choice([np.inf, -np.inf, np.nan, 1e30]))
    contract = contracts[int(rng.integers(0, len(contracts)))]
    tun = "contract=fma" if contract == "fma" else None
    o = OracleGmm(model, contract=contract)
    want, wbest = o.score(x)
    got, best = rasr_amd.GmmFeatureScorer(ctx, model, tuning=tun).score(x)
    ok = np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(best, wbest)
    # the byte form of the best-density matrix (amx_gmm_score_stats_u8_dev) and the best density / score of ONE mixture per frame
    # (amx_gmm_best_density_dev), both against the oracle
    import torch
    sc_u8 = rasr_amd.GmmFeatureScorer(ctx, model, tuning=tun)
    M = want.shape[1]
    ctx.use_torch_stream()
    xd = torch.from_numpy(x).cuda()
    s8 = torch.empty((T, M), dtype=torch.float32, device="cuda")
    b8 = torch.empty((T, M), dtype=torch.uint8, device="cuda")
    st = torch.empty((T,), dtype=torch.int32, device="cuda")
    cn = torch.zeros((M,), dtype=torch.int64, device="cuda")
    ss = torch.zeros((1,), dtype=torch.float64, device="cuda")
    ok_u8 = True
    if int(np.diff(model["mix_offsets"]).max()) <= 255:
        sc_u8.score_stats_dev(xd, T, s8, b8, st, cn, ss)
        torch.cuda.synchronize()
        ok_u8 = (np.array_equal(s8.cpu().numpy().view(np.uint32), want.view(np.uint32)) and
                 np.array_equal(b8.cpu().numpy(), np.where(wbest == 0xffffffff, 255, wbest).astype(np.uint8)))
    mix = rng.integers(0, M, T).astype(np.int32)
    bd = torch.empty((T,), dtype=torch.int32, device="cuda")
    sd = torch.empty((T,), dtype=torch.float32, device="cuda")
    sc_u8.best_density_dev(xd, T, torch.from_numpy(mix).cuda(), bd, sd)
    torch.cuda.synchronize()
    ok_bd = (np.array_equal(bd.cpu().numpy().astype(np.uint32), wbest[np.arange(T), mix]) and
             np.array_equal(sd.cpu().numpy().view(np.uint32), want[np.arange(T), mix].view(np.uint32)))
    ok = ok and ok_u8 and ok_bd
    sw, sb, _ = o.score_simd(x)
    sg, sgb = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="SIMD-diagonal-maximum").score(x)
    ok_simd = np.array_equal(sg.view(np.uint32), sw.view(np.uint32)) and np.array_equal(sgb, sb)
    if not (ok and ok_simd):
        bad += 1
        print("MISMATCH case %d: contract=%s kind=%s dim=%d pooled=%s seed=%d T=%d max=%s simd=%s" % (case, contract, kind, dim, pooled, seed, T, ok, ok_simd))
print("%d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
