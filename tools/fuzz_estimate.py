"""tools/fuzz_estimate.py [n_cases] [seed] -- random mixture-set topologies (shared means, shared covariances, densities used by several
mixtures or by none, empty mixtures) and random statistics through amx_gmm_estimate (host code, no GPU) against oracle/estimate.py,
bit for bit, with random estimator / splitter settings."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rasr_amd  # noqa: E402
from oracle import estimate as oe  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
bad = 0
for case in range(n_cases):
    dim = int(rng.integers(1, 9))
    n_mean, n_cov, n_dens, n_mix = (int(rng.integers(1, 12)) for _ in range(4))
    dens_mean = rng.integers(0, n_mean, n_dens).astype(np.uint32)
    dens_cov = rng.integers(0, n_cov, n_dens).astype(np.uint32)
    ks = rng.integers(0 if rng.integers(0, 4) == 0 else 1, 6, n_mix)
    off = np.concatenate([[0], np.cumsum(ks)]).astype(np.uint32)
    kd = np.concatenate([rng.choice(n_dens, k, replace=k > n_dens) for k in ks] + [np.zeros(0, np.int64)]).astype(np.uint32)
    # a mixture must not list a density twice (AbstractMixtureEstimator::addDensity requires it)
    ok = all(len(set(kd[off[m]:off[m + 1]])) == off[m + 1] - off[m] for m in range(n_mix))
    if not ok:
        continue
    nk = int(off[-1])
    model = dict(dim=dim, mix_offsets=off, dens_index=kd, log_weight=np.zeros(nk), dens_mean=dens_mean, dens_cov=dens_cov,
                 means=np.zeros((n_mean, dim), np.float32), variances=np.ones((n_cov, dim), np.float32))
    kw = rng.gamma(0.6, 20.0, nk) * (rng.random(nk) > 0.15)
    mw, cw = np.zeros(n_mean), np.zeros(n_cov)
    ms, cs = np.zeros((n_mean, dim)), np.zeros((n_cov, dim))
    for k in range(nk):
        d = kd[k]
        mu, sd = rng.standard_normal(dim) * 3, rng.uniform(0.2, 2.0, dim)
        mw[dens_mean[d]] += kw[k]
        cw[dens_cov[d]] += kw[k]
        ms[dens_mean[d]] += kw[k] * mu
        cs[dens_cov[d]] += kw[k] * (mu * mu + sd * sd)
    acc = np.concatenate([kw, mw, ms.reshape(-1), cw, cs.reshape(-1)])
    cfg = dict(min_observation_weight=float(rng.choice([0.0, 5.0, 30.0])), min_relative_weight=float(rng.choice([0.0, 0.1])),
               min_variance=float(rng.choice([0.0, 0.5])), normalize_mixture_weights=int(rng.integers(0, 2)), allow_zero_weights=1,
               split=int(rng.integers(0, 2)), split_min_mean_observation_weight=float(rng.choice([0.0, 20.0])),
               split_min_covariance_observation_weight=float(rng.choice([10.0, 3.4e38])), split_perturbation_weight=float(rng.choice([0.1, 1e5])),
               split_normalize_mixture_weights=int(rng.integers(0, 2)))
    topo = {k: model[k] for k in ("dim", "mix_offsets", "dens_index", "dens_mean", "dens_cov")}
    topo["n_mean"], topo["n_cov"] = n_mean, n_cov
    try:
        want = oe.estimate(topo, acc, **cfg)
    except Exception as e:
        print("oracle error", case, repr(e))
        bad += 1
        continue
    got = rasr_amd.gmm_estimate(model, acc, **cfg)
    same = all(np.array_equal(got[k], want[k]) for k in ("mix_offsets", "dens_index", "dens_mean", "dens_cov")) and \
        np.array_equal(got["log_weight"].view(np.uint64), np.asarray(want["log_weight"], np.float64).view(np.uint64)) and \
        np.array_equal(got["means"].view(np.uint32), want["means"].view(np.uint32)) and \
        np.array_equal(got["variances"].view(np.uint32), want["variances"].view(np.uint32))
    if not same:
        bad += 1
        print("MISMATCH case", case, "dim", dim, "n_mean", n_mean, "n_cov", n_cov, "n_dens", n_dens, "ks", list(ks), cfg)
print("%d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
