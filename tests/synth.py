"""Seeded synthetic inputs shared by tests/ and bench.py (recipes from SURVEY.md section 8d)."""
import numpy as np


def waveform(n_samples, seed, fs=16000.0):
    """x = round(3000*N(0,1) + 8000*sin(2 pi 440 t)*env) clipped to s16, cast to f32 (unscaled)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n_samples, dtype=np.float64) / fs
    env = 0.5 * (1.0 + np.sin(2 * np.pi * 0.7 * t + rng.uniform(0, 2 * np.pi)))
    x = 3000.0 * rng.standard_normal(n_samples) + 8000.0 * np.sin(2 * np.pi * 440.0 * t) * env
    return np.clip(np.rint(x), -32768, 32767).astype(np.float32)


def utterance_lengths(n_utt, seed, lo_s=5.0, hi_s=15.0, fs=16000.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.rint(rng.uniform(lo_s, hi_s, n_utt) * fs).astype(np.int64)


def gmm_cart(n_mix, k_lo, k_hi, dim, seed, pooled=True):
    """CART-style model: every mixture owns its densities (k_lo..k_hi each)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ks = rng.integers(k_lo, k_hi + 1, n_mix)
    off = np.zeros(n_mix + 1, np.uint32)
    off[1:] = np.cumsum(ks)
    nd = int(off[-1])
    means = rng.standard_normal((nd, dim)).astype(np.float32)
    n_cov = 1 if pooled else nd
    variances = rng.uniform(0.5, 2.0, (n_cov, dim)).astype(np.float32)
    logw = np.concatenate([np.log(rng.dirichlet(np.ones(k))) for k in ks]).astype(np.float64)
    return dict(dim=dim, mix_offsets=off, dens_index=np.arange(nd, dtype=np.uint32), log_weight=logw,
                dens_mean=np.arange(nd, dtype=np.uint32),
                dens_cov=(np.zeros(nd, np.uint32) if pooled else np.arange(nd, dtype=np.uint32)),
                means=means, variances=variances)


def gmm_tied(n_mix, n_dens, dim, seed, pooled=True, alpha=0.1, k_per_mix=None):
    """Tied-mixture model: n_dens shared densities; every mixture weights k_per_mix (default all) of them."""
    rng = np.random.Generator(np.random.PCG64(seed))
    k = n_dens if k_per_mix is None else k_per_mix
    means = rng.standard_normal((n_dens, dim)).astype(np.float32)
    n_cov = 1 if pooled else n_dens
    variances = rng.uniform(0.5, 2.0, (n_cov, dim)).astype(np.float32)
    off = (np.arange(n_mix + 1, dtype=np.uint64) * k).astype(np.uint32)
    if k == n_dens:
        idx = np.tile(np.arange(n_dens, dtype=np.uint32), n_mix)
    else:
        idx = np.concatenate([np.sort(rng.choice(n_dens, k, replace=False)) for _ in range(n_mix)]).astype(np.uint32)
    g = rng.gamma(alpha, 1.0, (n_mix, k)) + 1e-30
    logw = np.log(g / g.sum(axis=1, keepdims=True)).reshape(-1).astype(np.float64)
    return dict(dim=dim, mix_offsets=off, dens_index=idx, log_weight=logw,
                dens_mean=np.arange(n_dens, dtype=np.uint32),
                dens_cov=(np.zeros(n_dens, np.uint32) if pooled else np.arange(n_dens, dtype=np.uint32)),
                means=means, variances=variances)


def ffnn(dims, seed, act=1):
    """dims = [in, h1, ..., out]; weights ~ N(0, 1/sqrt(in)); log-prior = log softmax(N(0,1))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    Ws, bs, acts = [], [], []
    for l in range(len(dims) - 1):
        Ws.append((rng.standard_normal((dims[l + 1], dims[l])) / np.sqrt(dims[l])).astype(np.float32))
        bs.append((0.1 * rng.standard_normal(dims[l + 1])).astype(np.float32))
        acts.append(act if l < len(dims) - 2 else 0)
    z = rng.standard_normal(dims[-1])
    logp = (z - (np.log(np.sum(np.exp(z - z.max()))) + z.max())).astype(np.float32)
    return Ws, bs, acts, logp
