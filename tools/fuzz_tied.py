"""tools/fuzz_tied.py [n_cases] [seed] -- random tied models whose mixtures share one density list through the pruned exact scorer
(gmm_tied.hip, forced with amx_gmm_model.tuning tied_prune=1), the dense tile kernel (=0) and the adaptive default, against the oracle bit for
bit: density counts 1..9000, mixture counts that leave partial tiles, flat and peaked weights, zero-weight densities (a^ = +inf),
tiny variances (negative constants), huge constants, near ties and duplicates, frames with outliers / inf / NaN.
Not part of the test suite; run on a GPU box after touching gmm_tied.hip or the tied part of gmm.hip."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rasr_amd  # noqa: E402
from oracle import OracleGmm  # noqa: E402
from tests import synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
ctx = rasr_amd.Context(0)
bad = 0
for case in range(n_cases):
    dim = int(rng.choice([16, 24, 40, 7, 33]))
    nd = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 257, 1000, 4097, 9000], p=[.05, .05, .1, .1, .1, .1, .15, .15, .1, .07, .03]))
    n_mix = int(rng.choice([1, 5, 63, 64, 65, 130, 700]))
    if nd * n_mix > 1_500_000:
        n_mix = max(1, 1_500_000 // nd)
    seed = int(rng.integers(1, 1 << 30))
    alpha = float(rng.choice([0.05, 0.1, 1.0, 5.0]))
    model = synth.gmm_tied(n_mix, nd, dim, seed=seed, pooled=bool(rng.integers(0, 2)), alpha=alpha)
    twist = int(rng.integers(0, 9))
    if twist == 0:
        lw = model["log_weight"].copy()
        lw[rng.random(len(lw)) < 0.3] = -1.7976931348623157e+308      # zero-weight densities
        model["log_weight"] = lw
    elif twist == 1:
        model["variances"] = (model["variances"] * np.float32(10.0) ** rng.integers(-5, 2, model["variances"].shape)).astype(np.float32)
    elif twist == 2:
        model["means"] = (model["means"] * np.float32(50.0)).astype(np.float32)
    elif twist == 3:
        model["means"] = (model["means"][:1] + np.float32(1e-3) * model["means"]).astype(np.float32)        # near ties everywhere
    elif twist == 4:
        model["log_weight"] = model["log_weight"] + 2.0e5                                                  # huge constants
    elif twist == 5 and nd > 3:
        model["means"][1::2] = model["means"][0::2][:len(model["means"][1::2])]                             # duplicated densities
    elif twist == 6:
        model["log_weight"] = np.full_like(model["log_weight"], np.log(1.0 / nd))                           # flat weights
    T = int(rng.choice([1, 3, 4, 5, 63, 64, 65, 256, 300]))
    x = rng.standard_normal((T, dim)).astype(np.float32) * np.float32(rng.choice([0.3, 1.0, 3.0]))
    if rng.integers(0, 3) == 0:
        x[rng.integers(0, T)] *= np.float32(rng.choice([30.0, 1e4]))
    if rng.integers(0, 5) == 0:
        x[rng.integers(0, T), rng.integers(0, dim)] = np.float32(rng.choice([np.inf, -np.inf, np.nan, 1e30]))
    contract = ("off", "fma")[int(rng.integers(0, 2))]   # the reference's two arithmetics (amx_gmm_model.tuning contract=...)
    want, wbest = OracleGmm(model, contract=contract).score(x)
    status = []
    for mode in ("1", "0", None):
        sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning=",".join(i for i in (None if mode is None else "tied_prune=" + mode,
                                                                                   "contract=fma" if contract == "fma" else None) if i) or None)
        ok = True
        for _ in range(3 if mode is None else 1):        # the adaptive default: later calls see the statistics of earlier ones
            got, best = sc.score(x)
            ok = ok and np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(best, wbest)
        status.append(ok)
    if not all(status):
        bad += 1
        print("MISMATCH case %d: contract=%s dim=%d nd=%d n_mix=%d seed=%d alpha=%g twist=%d T=%d pruned/dense/adaptive=%s" %
              (case, contract, dim, nd, n_mix, seed, alpha, twist, T, status))
print("%d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
