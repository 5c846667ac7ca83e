"""oracle/acc_format.py -- byte-level restatement of RASR's binary mixture-set estimator ("accumulator") files.
TEST INFRASTRUCTURE; only tests/ may import this.

Independent second implementation (struct, whole file in memory) of what amx_gmm_accumulator_write / _read do, written
from the reference sources: Mm::AbstractMixtureSetEstimator::{writeHeader,write,read}
(src/Mm/AbstractMixtureSetEstimator.cc:404-508), Mm::VectorAccumulator::{write,read} (src/Mm/VectorAccumulator.hh:80-100),
Mm::GaussDensityEstimator::write (src/Mm/GaussDensityEstimator.cc:58-62), Mm::AbstractMixtureEstimator::write
(src/Mm/MixtureEstimator.cc:163-170), magic "MIXSET" (src/Mm/MixtureSetEstimator.hh:36-38), version 2
(src/Mm/AbstractMixtureSetEstimator.cc:75).  PARITY UNPINNED: the estimator classes need Core::Configuration (boost) and
the reference ships no estimator file.
"""
import struct

import numpy as np


def write(model, acc):
    """model: dict as for GmmFeatureScorer; acc: flat f64 [nk | n_mean | n_mean*dim | n_cov | n_cov*dim] -> bytes"""
    dim = int(model["dim"])
    off = np.asarray(model["mix_offsets"])
    nk, nm, nc = int(off[-1]), model["means"].shape[0], model["variances"].shape[0]
    nd = len(model["dens_mean"])
    a = np.asarray(acc, np.float64)
    mw, ms = a[nk:nk + nm], a[nk + nm:nk + nm + nm * dim].reshape(nm, dim)
    cw = a[nk + nm + nm * dim:nk + nm + nm * dim + nc]
    cs = a[nk + nm + nm * dim + nc:].reshape(nc, dim)
    out = [b"MIXSET\x00\x00", struct.pack("<II", 2, dim), struct.pack("<I", nm)]
    for i in range(nm):
        out += [struct.pack("<I", dim), ms[i].astype("<f8").tobytes(), struct.pack("<d", mw[i])]
    out.append(struct.pack("<I", nc))
    for i in range(nc):
        out += [struct.pack("<I", dim), cs[i].astype("<f8").tobytes(), struct.pack("<d", cw[i])]
    out.append(struct.pack("<I", nd))
    for d in range(nd):
        out.append(struct.pack("<II", int(model["dens_mean"][d]), int(model["dens_cov"][d])))
    out.append(struct.pack("<I", len(off) - 1))
    for m in range(len(off) - 1):
        out.append(struct.pack("<I", int(off[m + 1] - off[m])))
        for k in range(int(off[m]), int(off[m + 1])):
            out.append(struct.pack("<Id", int(model["dens_index"][k]), a[k]))
    return b"".join(out)


def read(b):
    """-> dict(version, dim, means [(sum, weight)], covariances [(sum, weight)], densities [(mean, cov)], mixtures [[(density, weight)]])"""
    assert b[:7] == b"MIXSET\x00"
    at = 8
    version, dim = struct.unpack_from("<II", b, at)
    at += 8

    def vecs():
        nonlocal at
        (n,) = struct.unpack_from("<I", b, at)
        at += 4
        res = []
        for _ in range(n):
            (size,) = struct.unpack_from("<I", b, at)
            at += 4
            v = np.frombuffer(b, "<f8", size, at).copy()
            at += 8 * size
            (w,) = struct.unpack_from("<d", b, at)
            at += 8
            res.append((v, w))
        return res
    means, covs = vecs(), vecs()
    (nd,) = struct.unpack_from("<I", b, at)
    at += 4
    dens = [struct.unpack_from("<II", b, at + 8 * i) for i in range(nd)]
    at += 8 * nd
    (nmix,) = struct.unpack_from("<I", b, at)
    at += 4
    mixtures = []
    for _ in range(nmix):
        (n,) = struct.unpack_from("<I", b, at)
        at += 4
        mixtures.append([struct.unpack_from("<Id", b, at + 12 * i) for i in range(n)])
        at += 12 * n
    assert at == len(b)
    return dict(version=version, dim=dim, means=means, covariances=covs, densities=dens, mixtures=mixtures)
