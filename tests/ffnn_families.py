"""Operand families for the f16 + MX-fp6 arithmetic (AMX_PREC_F16MX) that are NOT Gaussian: one e8m0 exponent per
32 k means a block with one outlier quantises the fp6 image of the other 31 values at 2^(E - 5) -- their cross terms then carry f16-grade
error only.  Each family returns (Ws, biases, activations, log_prior, inputs); tests/test_ffnn_f16mx_gpu.py holds every one of them to
north_star's bar (|d| <= 1e-4 |ref| + 1e-4 against the f64-accumulating oracle, best state identical), tools/fuzz_ffnn.py draws from them.
"""
import numpy as np

from tests import synth

FAMILIES = ("gaussian", "block-outliers-x", "block-outliers-w", "block-outliers-both", "lognormal-rows", "mfcc-context", "positive",
            "scaled-up", "scaled-down", "sparse-relu")


def _outliers(a, rng, lo=8, hi=12):
    """one element per 32-k block of every row multiplied by 2^u, u in [lo, hi]"""
    a = a.copy()
    rows, K = a.shape
    for b in range(0, K, 32):
        w = min(32, K - b)
        pos = rng.integers(0, w, rows)
        a[np.arange(rows), b + pos] *= np.exp2(rng.integers(lo, hi + 1, rows)).astype(np.float32)
    return a


def mfcc_context_windows(T, seed):
    """unnormalised MFCC-40 of synthetic audio, 11-frame context (c0 of tens beside c30 of hundredths) -- through the ORACLE front end, so
    that the family does not depend on the kernel under test"""
    from oracle import OracleMfcc
    m = OracleMfcc(n_ceps=40, filter_width=138.0)
    rows = []
    s = seed
    while sum(len(r) for r in rows) < T:
        c = m.run(synth.waveform(32000, seed=s))
        n = len(c)
        idx = np.clip(np.arange(n)[:, None] + np.arange(-5, 6)[None, :], 0, n - 1)
        rows.append(c[idx].reshape(n, 440))
        s += 1
    return np.concatenate(rows)[:T].astype(np.float32)


def make(family, dims, T, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    Ws, bs, acts, logp = synth.ffnn(dims, seed=seed + 1)
    x = rng.standard_normal((T, dims[0])).astype(np.float32)
    if family == "gaussian":
        pass
    elif family == "block-outliers-x":
        x = _outliers(x, rng)
    elif family == "block-outliers-w":
        Ws = [_outliers(w, rng, 4, 7) for w in Ws]   # (2^8 .. 2^12 on every layer's weights drives the activations out of the f16 range:
                                                     # that is the overflow contract's case, tested separately)
    elif family == "block-outliers-both":
        x = _outliers(x, rng, 6, 9)
        Ws = [_outliers(w, rng, 3, 5) for w in Ws]
    elif family == "lognormal-rows":
        Ws = [(np.sign(w) * np.exp(rng.normal(0.0, 1.5, w.shape)) * np.exp(rng.normal(0.0, 1.0, (w.shape[0], 1))) / np.sqrt(w.shape[1]) * 0.3)
              .astype(np.float32) for w in Ws]
    elif family == "mfcc-context":
        assert dims[0] == 440
        x = mfcc_context_windows(T, seed)
    elif family == "positive":
        x = np.abs(x)
        Ws = [(np.abs(w) * (2.0 / np.sqrt(w.shape[1]))).astype(np.float32) for w in Ws]   # nothing cancels: sums grow with K
        bs = [np.zeros_like(b) for b in bs]
    elif family == "scaled-up":
        x = (x * 300.0).astype(np.float32)
        Ws = [w.copy() for w in Ws]
    elif family == "scaled-down":
        x = (x * 1e-3).astype(np.float32)
    elif family == "sparse-relu":
        bs = [(b - 1.5).astype(np.float32) for b in bs]   # most hidden units off: blocks of zeros with a few live values
    else:
        raise ValueError(family)
    return Ws, bs, acts, logp, np.ascontiguousarray(x, dtype=np.float32)
