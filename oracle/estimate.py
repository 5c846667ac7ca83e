"""oracle/estimate.py -- restatement of RASR's mixture-set re-estimation (M-step) and splitting.  TEST INFRASTRUCTURE.

Only tests/ may import this (same rule as the rest of oracle/).  Plain Python / numpy scalars, small models only.

  estimate   Mm::AbstractMixtureSetEstimator::estimate            src/Mm/AbstractMixtureSetEstimator.cc:305-338
             removeDensitiesWithLowWeight                          src/Mm/MixtureEstimator.cc:64-82
             index maps (first-appearance numbering)               src/Mm/AbstractMixtureSetEstimator.cc:804-817
             AbstractMixtureEstimator::estimate, Mixture::addDensity / normalizeWeights
                                                                   src/Mm/MixtureEstimator.cc:118-129, src/Mm/Mixture.cc:63-74
             MeanEstimator::estimate, CovarianceEstimator::estimate, applyMinimumVariance
                                                                   src/Mm/GaussDensityEstimator.cc:148-158, 194-233
  split      Mm::MixtureSetSplitter                                src/Mm/MixtureSetSplitter.cc:38-123

The accumulator is the flat f64 buffer of amx_gmm_accumulator_size() doubles:
[sum K_m mixture-density weights][n_mean weights][n_mean x dim sums][n_cov weights][n_cov x dim sums of squares].

Pinning: the arithmetic that lives in the header-only Mm/Utilities.hh (logExpNorm, plusNormalizedSquare, normalizedMinus) is
checked against the reference's own templates in oracle/_ref/libref.so (tests/test_estimate.py, fixture
tests/golden/ref_estimate.json).  The control flow around it (density removal, renumbering, splitting) is PARITY UNPINNED:
the estimator classes need Core/Configuration.hh (boost) and the reference ships no estimator fixture.
One order is not defined by the reference: CovarianceEstimator::estimate walks an unordered_set of mean estimators keyed by
pointer; this restatement (and the product) walk them in mean-index order, results agree to f64 rounding of that sum.
"""
import math

import numpy as np

F32_EPS = np.float32(1.1920929e-07)   # Core::Type<f32>::epsilon
F64_MIN = -1.7976931348623157e+308    # Core::Type<f64>::min

DEFAULTS = dict(min_observation_weight=5.0, min_relative_weight=0.0, min_variance=0.0, normalize_mixture_weights=True,
                allow_zero_weights=False, split=False, split_min_mean_observation_weight=20.0,
                split_min_covariance_observation_weight=float(np.finfo(np.float32).max), split_perturbation_weight=0.1,
                split_normalize_mixture_weights=False)


def log_exp_norm(v):
    """Mm/Utilities.hh:43-51: first maximum, sum of exp over the others, log1p"""
    mx = 0
    for i in range(1, len(v)):
        if v[mx] < v[i]:
            mx = i
    r = 0.0
    for i, x in enumerate(v):
        if i != mx:
            r += math.exp(x - v[mx])
    return math.log1p(r) + v[mx]


def plus_normalized_square(sum_, y, n):
    """Mm/Utilities.hh:155-170: x + y*y / n, element-wise in f64"""
    return np.asarray(sum_, np.float64) + np.asarray(y, np.float64) * np.asarray(y, np.float64) / np.float64(n)


def normalized_minus(x, y, n):
    """Mm/Utilities.hh:95-106: (x - y) / n in f64, stored as VarianceType (f32)"""
    return ((np.asarray(x, np.float64) - np.asarray(y, np.float64)) / np.float64(n)).astype(np.float32)


def estimate(model, acc, **cfg):
    """model: dict(dim, mix_offsets, dens_index, dens_mean, dens_cov) (n_mean / n_cov inferred); acc: flat f64.
    -> dict(dim, mix_offsets, dens_index, log_weight, dens_mean, dens_cov, means, variances) + zero_weight_mixtures"""
    c = dict(DEFAULTS)
    c.update(cfg)
    dim = int(model["dim"])
    off = [int(v) for v in model["mix_offsets"]]
    kd = [int(v) for v in model["dens_index"]]
    dmean = [int(v) for v in model["dens_mean"]]
    dcov = [int(v) for v in model["dens_cov"]]
    n_mix, nk, n_mean, n_cov = len(off) - 1, off[-1], int(model["n_mean"]), int(model["n_cov"])
    acc = np.asarray(acc, np.float64)
    kw = acc[:nk]
    mw = acc[nk:nk + n_mean]
    ms = acc[nk + n_mean:nk + n_mean + n_mean * dim].reshape(n_mean, dim)
    o = nk + n_mean + n_mean * dim
    cw = acc[o:o + n_cov]
    cs = acc[o + n_cov:o + n_cov + n_cov * dim].reshape(n_cov, dim)

    # checkEventsWithZeroWeight (:422-431): a mixture without any weight is a critical error unless allowed
    zero = [m for m in range(n_mix) if float(np.sum(kw[off[m]:off[m + 1]])) == 0.0 and off[m + 1] > off[m]]
    if zero and not c["allow_zero_weights"]:
        raise ValueError("Mixture %d has zero weight." % zero[0])

    # CovarianceToMeanSetMap over ALL densities reachable from the mixtures, before the removal (:309-313)
    cov_means = {}
    for k in kd:
        cov_means.setdefault(dcov[k], set()).add(dmean[k])

    # removeDensitiesWithLowWeight per mixture
    mixtures = []
    for m in range(n_mix):
        ent = [(kd[k], float(kw[k])) for k in range(off[m], off[m + 1])]
        if ent:
            dmax = 0
            for i in range(1, len(ent)):
                if ent[i][1] > ent[dmax][1]:
                    dmax = i
            total = 0.0
            for _, w in ent:
                total += w
            min_w = max(c["min_observation_weight"], total * c["min_relative_weight"])
            i = 0
            while i < len(ent):
                if not (ent[i][1] >= min_w) and i != dmax:
                    del ent[i]
                    if dmax > i:
                        dmax -= 1
                else:
                    i += 1
        mixtures.append(ent)

    # index maps: first appearance while walking mixtures, densities in mixture order
    dmap, mmap, cmap = {}, {}, {}
    for ent in mixtures:
        for d, _ in ent:
            mmap.setdefault(dmean[d], len(mmap))
            cmap.setdefault(dcov[d], len(cmap))
            dmap.setdefault(d, len(dmap))

    out_off, out_kd, out_lw = [0], [], []
    for ent in mixtures:
        lw = [math.log(w) if w > 0 else F64_MIN for _, w in ent]
        if c["normalize_mixture_weights"] and lw:
            norm = log_exp_norm(lw)
            lw = [x - norm for x in lw]
        out_kd += [dmap[d] for d, _ in ent]
        out_lw += lw
        out_off.append(len(out_kd))
    inv = lambda mp: [k for k, _ in sorted(mp.items(), key=lambda kv: kv[1])]
    out_dmean = [mmap[dmean[d]] for d in inv(dmap)]
    out_dcov = [cmap[dcov[d]] for d in inv(dmap)]

    means = np.zeros((len(mmap), dim), np.float32)
    mean_w = []
    for new, old in enumerate(inv(mmap)):
        mean_w.append(float(mw[old]))
        if mw[old] != 0:
            means[new] = (ms[old] / mw[old]).astype(np.float32)      # f64 division, stored as MeanType
    variances = np.ones((len(cmap), dim), np.float32)                 # DiagonalCovariance(dim): ones
    cov_w = []
    min_var = np.float32(c["min_variance"])
    for new, old in enumerate(inv(cmap)):
        cov_w.append(float(cw[old]))
        if cw[old] == 0:
            continue
        wmss = np.zeros(dim, np.float64)
        for mi in sorted(cov_means[old]):
            if mw[mi] > 0:
                wmss = plus_normalized_square(wmss, ms[mi], mw[mi])
        v = normalized_minus(cs[old], wmss, cw[old])
        if min_var != 0:
            v = np.where(v < min_var, min_var, v)
        variances[new] = v

    res = dict(dim=dim, mix_offsets=out_off, dens_index=out_kd, log_weight=out_lw, dens_mean=out_dmean, dens_cov=out_dcov,
               means=means, variances=variances, zero_weight_mixtures=zero)
    if c["split"]:
        _split(res, mean_w, cov_w, c)
    res["mix_offsets"] = np.asarray(res["mix_offsets"], np.uint32)
    res["dens_index"] = np.asarray(res["dens_index"], np.uint32)
    res["dens_mean"] = np.asarray(res["dens_mean"], np.uint32)
    res["dens_cov"] = np.asarray(res["dens_cov"], np.uint32)
    res["log_weight"] = np.asarray(res["log_weight"], np.float64)
    return res


def _split(r, mean_w, cov_w, c):
    dim = r["dim"]
    means = [row.copy() for row in r["means"]]
    variances = [row.copy() for row in r["variances"]]
    dmean, dcov = list(r["dens_mean"]), list(r["dens_cov"])
    n_dens, n_mean0, n_cov0 = len(dmean), len(means), len(variances)
    # splitMeans (:49-65): ONE pass over the densities; a mean shared by several densities is split once per density
    split_mean = list(range(n_mean0))
    pw = float(c["split_perturbation_weight"])
    for d in range(n_dens):
        mi, ci = dmean[d], dcov[d]
        pert = np.array([np.float32(float(np.sqrt(np.float32(v))) * pw * float(F32_EPS)) for v in variances[ci]], np.float32)
        if mean_w[mi] > c["split_min_mean_observation_weight"]:
            means.append((means[mi] - pert).astype(np.float32))
            means[mi] = (means[mi] + pert).astype(np.float32)
            split_mean[mi] = len(means) - 1
        else:
            split_mean[mi] = mi
    split_cov = list(range(n_cov0))
    for ci in range(n_cov0):
        if cov_w[ci] > c["split_min_covariance_observation_weight"]:
            variances.append(variances[ci].copy())
            split_cov[ci] = len(variances) - 1
    split_dens = list(range(n_dens))
    for d in range(n_dens):
        sm, sc = split_mean[dmean[d]], split_cov[dcov[d]]
        if sm != dmean[d] or sc != dcov[d]:
            dmean.append(sm)
            dcov.append(sc)
            split_dens[d] = len(dmean) - 1
    off, kd, lw = r["mix_offsets"], r["dens_index"], r["log_weight"]
    n_off, n_kd, n_lw = [0], [], []
    for m in range(len(off) - 1):
        ent_d = list(kd[off[m]:off[m + 1]])
        ent_w = list(lw[off[m]:off[m + 1]])
        for j in range(off[m + 1] - off[m]):
            sd = split_dens[ent_d[j]]
            if sd != ent_d[j]:
                ent_d.append(sd)
                ent_w.append(ent_w[j])
        if c["split_normalize_mixture_weights"] and ent_w:
            norm = log_exp_norm(ent_w)
            ent_w = [x - norm for x in ent_w]
        n_kd += ent_d
        n_lw += ent_w
        n_off.append(len(n_kd))
    r.update(mix_offsets=n_off, dens_index=n_kd, log_weight=n_lw, dens_mean=dmean, dens_cov=dcov,
             means=np.array(means, np.float32).reshape(-1, dim), variances=np.array(variances, np.float32).reshape(-1, dim))
