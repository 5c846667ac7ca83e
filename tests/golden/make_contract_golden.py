"""Regenerates tests/golden/ref_contract.npz and profiles/r05/contract_pins.json.  Run in the build container only:

    python tests/golden/make_contract_golden.py

The reference has two arithmetics (cmake_resources/CompileOptions.cmake:21-48): built with -DMARCH=x86-64 it rounds every operation
once; its DEFAULT configuration (-march=native, GCC's -ffp-contract=fast) on an FMA host fuses a product whose only use is an
addition into one fused multiply-add.  oracle/ref/Makefile compiles the reference both ways (libref.so: -msse3; libref_native.so:
-msse3 -march=native) -- whole translation units where they build, and FUNCTION-TEXT pins (oracle/ref/extract_fn.py) for
Mm::GaussDiagonalMaximumFeatureScorer::distance, Signal::Regression, Signal::FilterBank::Filter::apply,
Signal::HammingWindowFunction::init, Mm::BatchFloatFeatureScorer::fillScoreCacheTpl, Signal::Preemphasis and
Signal::FilterBank::FilterBuilder (one filter: interval and weights).

ref_contract.npz: seeded INPUTS and the OUTPUTS OF THE REFERENCE in both flavours ("<pin>_off", "<pin>_fma") for the
contraction-sensitive pins; tests/test_contract.py holds both oracle libraries to them bit for bit, everywhere (no reference tree
needed).  contract_pins.json: for EVERY pin of libref, how many of the tried inputs give different bits in the two flavours.
Only data is stored -- no reference source text.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.binding import load_ref  # noqa: E402

import ctypes as C  # noqa: E402


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def ndiff(a, b):
    return int(np.count_nonzero(bits(a) != bits(b)))


def main():
    R = {c: load_ref(c) for c in ("off", "fma")}
    assert all(R.values()), "both flavours of oracle/_ref are needed (make -C oracle/ref)"
    rng = np.random.default_rng(20260930)
    gold, report = {}, {}

    # ---- a13 / a14: the distance (function-text pin)
    dims = [40, 39, 33, 32, 24, 16, 7, 3, 45]
    n = 250
    for dim in dims:
        x = rng.standard_normal((n, dim)).astype(np.float32)
        mu = rng.standard_normal((n, dim)).astype(np.float32)
        isr = rng.uniform(0.5, 2.0, (n, dim)).astype(np.float32)
        # a few large-magnitude rows: products far from 1
        x[:20] *= 100.0
        mu[20:40] *= 1e-3
        gold["dist_x_%d" % dim], gold["dist_mu_%d" % dim], gold["dist_isr_%d" % dim] = x, mu, isr
        for c in R:
            gold["dist_%d_%s" % (dim, c)] = np.array([R[c].ref_gdm_distance(x[i], mu[i], isr[i], dim) for i in range(n)], np.float32)
    tot = sum(ndiff(gold["dist_%d_off" % d], gold["dist_%d_fma" % d]) for d in dims)
    report["Mm::GaussDiagonalMaximumFeatureScorer::distance (function text, GaussDiagonalMaximumFeatureScorer.cc:144-218)"] = dict(
        tried=n * len(dims), differ=tot, fma_sites="vfmadd231ps (sum += df * df), vfmadd231ss (tail)")

    # ---- f1: regression (function text)
    for order in (1, 2):
        for right in (1, 2, 3):
            nin, dim = 2 * right + 1, 45
            w = (rng.standard_normal((60, nin, dim)) * 10).astype(np.float32)
            gold["reg_in_%d_%d" % (order, right)] = w
            for c in R:
                out = np.zeros((60, dim), np.float32)
                for i in range(60):
                    R[c].ref_regression(order, w[i].reshape(-1), nin, dim, out[i])
                gold["reg_%d_%d_%s" % (order, right, c)] = out
    report["Signal::Regression::regressFirstOrder / regressSecondOrder (function text, Regression.cc:24-65)"] = dict(
        tried=6 * 60 * 45, differ=sum(ndiff(gold["reg_%d_%d_off" % (o, r)], gold["reg_%d_%d_fma" % (o, r)]) for o in (1, 2) for r in (1, 2, 3)),
        fma_sites="first order: out += dt * f, tm += dt * dt; second order: ns += (dt^3) * dt, ns = tm * tm - n * ns (vfmsub), out += f * tm, "
                  "out -= (f dt dt) * n (vfnmadd); NOT tm += dt * dt there (the product has a second use)")

    # ---- a7: one filter of the bank (function text)
    nb = 257
    amp = np.abs(rng.standard_normal((200, nb)) * 50).astype(np.float32)
    start = rng.integers(0, 200, 200).astype(np.int32)
    end = (start + rng.integers(1, 57, 200)).astype(np.int32)
    wts = rng.uniform(0, 1, (200, 56)).astype(np.float32)
    gold["fb_amp"], gold["fb_start"], gold["fb_end"], gold["fb_w"] = amp, start, end, wts
    for c in R:
        gold["fb_%s" % c] = np.array([R[c].ref_filter_apply(amp[i], nb, int(start[i]), int(end[i]), wts[i]) for i in range(200)], np.float32)
    report["Signal::FilterBank::Filter::apply (function text, Filterbank.cc:27-50,65-71)"] = dict(
        tried=200, differ=ndiff(gold["fb_off"], gold["fb_fma"]), fma_sites="vfmadd231ss (result += in[f] * weights_[f - start_])")

    # ---- a10 / f1: Math::Matrix x Math::Vector (whole TU headers)
    M = rng.standard_normal((40, 40)).astype(np.float32)
    v = (rng.standard_normal((100, 40)) * 3).astype(np.float32)
    gold["mv_M"], gold["mv_v"] = M, v
    for c in R:
        out = np.zeros((100, 40), np.float32)
        for i in range(100):
            R[c].ref_matrix_vector(M.ctypes.data, 40, 40, v[i].ctypes.data, out[i].ctypes.data)
        gold["mv_%s" % c] = out
    report["Math::Matrix<f32> * Math::Vector<f32> (Matrix.hh:485-494, Vector.hh:94-101): cosine transform, signal-matrix-multiplication"] = dict(
        tried=4000, differ=ndiff(gold["mv_off"], gold["mv_fma"]), fma_sites="vfmadd231ss (result += a[i] * b[i])")

    # ---- a15: gaussLogNormFactor (f64), as the scorer uses it: narrowed to f32
    tot = tot32 = 0
    for dim in (40, 39, 33, 24, 13):
        var = rng.uniform(0.05, 20.0, (120, dim)).astype(np.float32)
        gold["ln_var_%d" % dim] = var
        for c in R:
            gold["ln_%d_%s" % (dim, c)] = np.array([R[c].ref_gauss_log_norm_factor(var[i], dim) for i in range(120)], np.float64)
        tot += ndiff(gold["ln_%d_off" % dim], gold["ln_%d_fma" % dim])
        tot32 += ndiff(gold["ln_%d_off" % dim].astype(np.float32), gold["ln_%d_fma" % dim].astype(np.float32))
    report["Mm::gaussLogNormFactor (Utilities.hh:70-75), f64, dims 40 39 33 24 13"] = dict(
        tried=600, differ=tot, differ_after_f32=tot32, fma_sites="vfmadd (N * log(2 pi) + logNorm); 40 * log(2 pi) happens to be exact in f64")

    # ---- pins that turn out flag-INSENSITIVE on everything tried: recorded, no fixture needed
    def fft_pair(nfl, real):
        d = 0
        for _ in range(300):
            a = (rng.standard_normal(nfl) * 3000).astype(np.float32)
            o = []
            for c in R:
                b = a.copy()
                (R[c].ref_fft_real if real else R[c].ref_fft_complex)(b, nfl)
                o.append(b)
            d += ndiff(o[0], o[1])
        return d
    report["Math::FastFourierTransform::transformReal (FastFourierTransform.cc), 512 points, 300 frames"] = dict(
        tried=300 * 512, differ=fft_pair(512, True), fma_sites="14 f64 fused operations (twiddle recurrence, butterflies' f64 products, real split); "
                                                               "the f32 results hide them")
    report["Math::FastFourierTransform::transform, 1024 floats, 300 frames"] = dict(tried=300 * 1024, differ=fft_pair(1024, False))

    from oracle.binding import ref_levinson
    import oracle.binding as B
    d = tried = 0
    for _ in range(300):
        nac = 13
        x = rng.standard_normal(400)
        Rv = np.array([np.dot(x[:400 - k], x[k:]) for k in range(nac)], np.float32)
        outs = []
        for c in R:
            gain = C.c_float(0)
            a = np.zeros(nac, np.float32)
            R[c].ref_levinson.restype = C.c_int
            R[c].ref_levinson.argtypes = [B.f32p, C.c_int, C.POINTER(C.c_float), B.f32p]
            ok = R[c].ref_levinson(Rv, nac, C.byref(gain), a)
            outs.append(np.concatenate([[np.float32(gain.value)], a]).astype(np.float32))
        d += ndiff(outs[0], outs[1])
        tried += nac + 1
    report["Math::LevinsonLeastSquares (LevinsonLse.cc), order 12, 300 autocorrelations (MF-PLP / PLP, compared at 1e-4)"] = dict(
        tried=tried, differ=d, fma_sites="4 f64 fused operations")

    fns = [("ref_mel", 1), ("ref_mel_derivative", 1), ("ref_mel_inverse", 1), ("ref_bark", 1), ("ref_bark_derivative", 1), ("ref_bark_inverse", 1)]
    for name, _ in fns:
        xs = rng.uniform(0, 8000, 2000) if "inverse" not in name else rng.uniform(0, 20, 2000)
        a = np.array([getattr(R["off"], name)(float(t)) for t in xs])
        b = np.array([getattr(R["fma"], name)(float(t)) for t in xs])
        report["%s (f64, filter-bank construction)" % name] = dict(tried=2000, differ=ndiff(a, b))
    a = np.array([R["off"].ref_inverse_square_root(float(t)) for t in gold["ln_var_40"][:50].reshape(-1)], np.float32)
    b = np.array([R["fma"].ref_inverse_square_root(float(t)) for t in gold["ln_var_40"][:50].reshape(-1)], np.float32)
    report["Mm::inverseSquareRoot<f32> (Utilities.hh:86-91)"] = dict(tried=2000, differ=ndiff(a, b))

    # ---- a3: the Hamming table (function text) -- flag-insensitive for every length tried; the fixture keeps the lengths the flow
    # files use (25 ms / 20 ms at 16 kHz and 8 kHz) and a few odd ones
    rng2 = np.random.default_rng(20260931)   # (a second generator: the draws above keep their values)
    lens = [2, 3, 160, 200, 320, 400, 401, 512, 1001]
    d = 0
    for n in range(2, 4097):
        a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
        R["off"].ref_hamming_window(n, a)
        R["fma"].ref_hamming_window(n, b)
        d += ndiff(a, b)
        if n in lens:
            gold["hamming_%d" % n] = a
    report["Signal::HammingWindowFunction::init (function text, WindowFunction.cc:92-101), every length 2 .. 4096"] = dict(
        tried=sum(range(2, 4097)), differ=d, fma_sites="vfnmadd132sd (0.54 - 0.46 * cos()), f64; the f32 table hides it")

    # ---- a18: BatchFloatFeatureScorer::fillScoreCacheTpl (function text): pre-scaled means / features of one mixture, with the
    # non-finite cases that separate _mm_min_ps(score, s) from min(s, score)
    cases = []
    for dim in (40, 39, 33, 16, 8, 45, 3):
        pdim = (dim + 7) // 8 * 8
        for trial in range(12):
            nk, T = int(rng2.integers(1, 20)), int(rng2.integers(1, 10))
            ms = np.zeros((nk, pdim), np.float32)
            ms[:, :dim] = rng2.standard_normal((nk, dim)) * rng2.uniform(0.1, 30)
            xs = np.zeros((T, pdim), np.float32)
            xs[:, :dim] = rng2.standard_normal((T, dim)) * rng2.uniform(0.1, 30)
            cst = (rng2.standard_normal(nk) * 20 + 60).astype(np.float32)
            if trial % 6 == 3:
                xs[0, 0] = np.nan                      # every density's sum is NaN: the score is NaN
            if trial % 6 == 4 and nk > 1:
                ms[0, 1], xs[T - 1, 1] = np.inf, np.inf  # first density NaN (inf - inf), the others +inf: NaN, then replaced by +inf
            if trial % 6 == 5:
                ms[nk - 1, 2], xs[0, 2] = np.inf, np.inf  # the LAST density NaN: the score ends as NaN
            cases.append((ms, cst, xs))
    gold["bf_n"] = np.array([len(cases)])
    tot = dif = 0
    for i, (ms, cst, xs) in enumerate(cases):
        gold["bf_ms_%d" % i], gold["bf_cst_%d" % i], gold["bf_xs_%d" % i] = ms, cst, xs
        for c in R:
            out = np.zeros(len(xs), np.float32)
            R[c].ref_batch_float_fill(ms.reshape(-1), cst, len(ms), xs.reshape(-1), len(xs), ms.shape[1], out)
            gold["bf_%d_%s" % (i, c)] = out
        tot += len(xs)
        dif += ndiff(gold["bf_%d_off" % i], gold["bf_%d_fma" % i])
    report["Mm::BatchFloatFeatureScorer::fillScoreCacheTpl (function text, BatchFeatureScorer.cc:207-253)"] = dict(
        tried=tot, differ=dif, fma_sites="2 x vfmadd231ps (s = _mm_add_ps(s, _mm_mul_ps(x, x)): intrinsics are vector arithmetic to GCC)")

    # ---- a1: Signal::Preemphasis (function text; the whole class): 4096-sample blocks with contiguous time stamps, alpha = 1 (mfcc.flow:
    # no product) and alpha != 1 (v[i] -= alpha * previous: vfnmadd132ss in the default build), and a restart behind a gap
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
    for c in R:
        R[c].ref_preemphasis.argtypes = [C.c_float, C.c_double, f32p, C.c_long, C.c_int, C.c_int, f32p]
    x = (rng2.standard_normal(9000) * 3000).astype(np.float32)
    gold["pre_x"] = x
    tot = dif = 0
    for name, alpha, gap in (("1", 1.0, -1), ("097", 0.97, -1), ("05", 0.5, -1), ("097_gap", 0.97, 1)):
        for c in R:
            out = np.zeros_like(x)
            R[c].ref_preemphasis(alpha, 16000.0, x, len(x), 4096, gap, out)
            gold["pre_%s_%s" % (name, c)] = out
        tot += len(x)
        dif += ndiff(gold["pre_%s_off" % name], gold["pre_%s_fma" % name])
    report["Signal::Preemphasis (function text, Preemphasis.cc:23-74), alpha 1 / 0.97 / 0.5, blocks of 4096, one restart"] = dict(
        tried=tot, differ=dif, fma_sites="vfnmadd132ss (v[i] -= alpha * previous); none for alpha = 1, the flow files' value")

    # ---- a7: FilterBuilder::create for one filter (function text): interval and weights, triangular / trapeze x mel / bark x
    # differential unit, centres on and off the grid of the discrete axis, filters that stick out at either end and ones the builder refuses
    import math
    ip = C.POINTER(C.c_int)
    for c in R:
        R[c].ref_filter_build.restype = C.c_int
        R[c].ref_filter_build.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, ip, ip, f32p, C.c_int]
    params, outs = [], {c: [] for c in R}
    for trial in range(400):
        typ, warp, diff = int(rng2.integers(0, 2)), int(rng2.integers(0, 2)), int(rng2.integers(0, 2))
        fs, N = float(rng2.choice([8000., 16000., 11025.])), int(rng2.choice([256, 512, 1024]))
        B, d2c = N // 2 + 1, 1.0 / (N / fs)
        wv = (lambda f: 2595.0 * math.log10(1 + f / 700.0)) if warp == 0 else (lambda f: 6.0 * math.asinh(f / 600.0))
        fmaxw = wv(d2c * (B - 1))
        width = float(rng2.uniform(0.5, 400.0)) if warp == 0 else float(rng2.uniform(0.5, 6.0))
        center = float(rng2.uniform(-0.2 * fmaxw, 1.2 * fmaxw)) if trial % 5 else wv(d2c * int(rng2.integers(0, B)))
        params.append([typ, warp, center, width, 0.0, fmaxw, d2c, diff])
        for c in R:
            st, en, w = C.c_int(0), C.c_int(0), np.zeros(600, np.float32)
            n = R[c].ref_filter_build(typ, warp, center, width, 0.0, fmaxw, d2c, diff, C.byref(st), C.byref(en), w, 600)
            outs[c].append((n, st.value if n >= 0 else -1, en.value if n >= 0 else -1, w[:max(n, 0)].copy()))
    gold["fbb_params"] = np.array(params, np.float64)
    dif = 0
    for c in R:
        gold["fbb_n_%s" % c] = np.array([o[0] for o in outs[c]], np.int32)
        gold["fbb_start_%s" % c] = np.array([o[1] for o in outs[c]], np.int32)
        gold["fbb_end_%s" % c] = np.array([o[2] for o in outs[c]], np.int32)
        gold["fbb_w_%s" % c] = np.concatenate([o[3] for o in outs[c]]).astype(np.float32)
    same_shape = all(np.array_equal(gold["fbb_%s_off" % k], gold["fbb_%s_fma" % k]) for k in ("n", "start", "end"))
    dif = ndiff(gold["fbb_w_off"], gold["fbb_w_fma"]) if same_shape else -1
    report["Signal::FilterBank::FilterBuilder::create / setStart / setEnd / setWeights + triangular / trapeze weight (function text, "
           "Filterbank.cc:144-217,236-244,268-281,691-694), 400 filters"] = dict(
        tried=int(len(gold["fbb_w_off"])), differ=dif, refused_by_the_builder=int(np.count_nonzero(gold["fbb_n_off"] < 0)),
        fma_sites="none that reaches a result (the weight is one f32 x f64 product, the triangle a division)")

    # ---- a7: the boundary of a bank (function text for Boundary::init / setSpacing / postprocessNumberOfFilters,
    # IncludeBoundary::getNumberOfFilters, StretchToCover::init; the one-line centre formulas are retyped class bodies): number of filters,
    # final width and spacing, centres.  The default build fuses spacing * (n - 1) + width, the two products of the stretch-to-cover
    # centre and max - (1 - ncp) * width
    dp = C.POINTER(C.c_double)
    f64p = np.ctypeslib.ndpointer(np.float64, flags="C")
    for c in R:
        R[c].ref_filter_boundary.restype = C.c_int
        R[c].ref_filter_boundary.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, f64p, C.c_int]
    bpar, bout = [], {c: [] for c in R}
    for trial in range(300):
        typ, ncp = int(rng2.integers(0, 3)), float(rng2.choice([0.5, 2.5 / 3.8]))
        fmax = float(rng2.uniform(5, 3000))
        width = float(rng2.uniform(0.01, 1.2)) * fmax if trial % 3 else fmax / float(rng2.integers(1, 60))
        spacing = 0.0 if trial % 2 else float(rng2.uniform(0.005, 0.6)) * fmax
        if trial == 0:
            typ, ncp, fmax, width, spacing = 0, 0.5, 2595.0 * math.log10(1 + 8000.0 / 700.0), 268.258, 0.0   # mfcc.flow's bank at 16 kHz
        bpar.append([typ, width, spacing, ncp, 0.0, fmax])
        for c in R:
            w, sp, cen = C.c_double(0), C.c_double(0), np.zeros(256)
            n = R[c].ref_filter_boundary(typ, width, spacing, ncp, 0.0, fmax, C.byref(w), C.byref(sp), cen, 256)
            bout[c].append((n, w.value, sp.value, cen[:max(0, min(n, 256))].copy()))
    gold["fbd_params"] = np.array(bpar, np.float64)
    for c in R:
        gold["fbd_n_%s" % c] = np.array([o[0] for o in bout[c]], np.int32)
        gold["fbd_ws_%s" % c] = np.array([[o[1], o[2]] for o in bout[c]], np.float64)
        gold["fbd_centers_%s" % c] = np.concatenate([o[3] for o in bout[c]])
    report["Signal::FilterBank::Boundary (function text, Filterbank.cc:428-470,495-501,546-567; centre formulas retyped), 300 banks"] = dict(
        tried=int(len(gold["fbd_centers_off"]) + 2 * 300), differ=ndiff(gold["fbd_centers_off"], gold["fbd_centers_fma"]) + ndiff(gold["fbd_ws_off"], gold["fbd_ws_fma"]),
        filter_counts_differ=int(np.count_nonzero(gold["fbd_n_off"] != gold["fbd_n_fma"])),
        fma_sites="vfmadd132sd (coverage: spacing * (n - 1) + width), 2 x vfmadd (stretch-to-cover centre), vfnmadd132sd (include-boundary: "
                  "max - (1 - ncp) * width); in FilterBuilder: vfnmadd231sd (setStart), vfmadd132sd (setEnd); DerivedArcSinh (bark derivative): vfmadd132sd")

    # ---- a10: Signal::CosineTransform (function text, the whole class): the two tables and apply with and without the division by N
    for c in R:
        R[c].ref_cosine_transform.restype = None
        R[c].ref_cosine_transform.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p]
    ct_cases = [(0, 40, 40, 0), (0, 20, 16, 0), (1, 22, 20, 1), (1, 17, 13, 1), (0, 33, 12, 1), (1, 2, 2, 0), (0, 7, 7, 0)]
    gold["ct_cases"] = np.array(ct_cases, np.int32)
    dt = do = nt = no = 0
    for i, (np1, n_in, n_out, norm) in enumerate(ct_cases):
        x = (rng2.standard_normal((8, n_in)) * 5).astype(np.float32)
        gold["ct_in_%d" % i] = x
        for c in R:
            out, tab = np.zeros((8, n_out), np.float32), np.zeros(n_out * n_in, np.float32)
            for r in range(8):
                R[c].ref_cosine_transform(np1, n_in, n_out, norm, x[r], out[r], tab)
            gold["ct_out_%d_%s" % (i, c)], gold["ct_tab_%d_%s" % (i, c)] = out, tab
        dt += ndiff(gold["ct_tab_%d_off" % i], gold["ct_tab_%d_fma" % i])
        do += ndiff(gold["ct_out_%d_off" % i], gold["ct_out_%d_fma" % i])
        nt += n_out * n_in
        no += 8 * n_out
    report["Signal::CosineTransform tables (function text, CosineTransform.cc:20-83)"] = dict(tried=nt, differ=dt)
    report["Signal::CosineTransform::apply (the same pin)"] = dict(tried=no, differ=do, fma_sites="vfmadd231ss (Math::Vector's dot product)")

    # ---- f4 (MF-PLP / PLP): Signal::autoregressionToCepstrum (function text)
    for c in R:
        R[c].ref_ar_to_cepstrum.restype = None
        R[c].ref_ar_to_cepstrum.argtypes = [C.c_float, f32p, C.c_int, f32p, C.c_int]
    A = (rng2.standard_normal((120, 20)) * 0.5).astype(np.float32)
    G = rng2.uniform(0.01, 50, 120).astype(np.float32)
    gold["arc_a"], gold["arc_gain"] = A, G
    for c in R:
        out = np.zeros((120, 16), np.float32)
        for i in range(120):
            R[c].ref_ar_to_cepstrum(float(G[i]), A[i], 20, out[i], 16)
        gold["arc_%s" % c] = out
    report["Signal::autoregressionToCepstrum (function text, AutoregressionToCepstrum.cc:21-36), order 20 -> 16 cepstra"] = dict(
        tried=120 * 16, differ=ndiff(gold["arc_off"], gold["arc_fma"]), fma_sites="vfmadd132ss (c[n] += ((n - k) * c[n - k]) * a[k - 1])")

    # ---- f4: signal-gammatone (function text: WarpingFunction + GammaTone, design and cascade), blocks of 256 samples
    for c in R:
        R[c].ref_gammatone.restype = C.c_int
        R[c].ref_gammatone.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_char_p, f32p,
                                       C.c_long, C.c_int, f32p, f32p, f32p]
    # (sample rate, cascade, minfreq, maxfreq, q, channels, cfmode, warp-freqbreak, warping-factor)
    gt_cases = [(16000.0, 4, 100.0, 6000.0, 9.264491981582191, 12, 0, 6600.0, "1"), (8000.0, 3, 80.0, 3800.0, 9.264491981582191, 10, 1, 3300.0, "1"),
                (16000.0, 4, 100.0, 7500.0, 7.5, 16, 0, 6600.0, "0.9"), (16000.0, 1, 50.0, 6000.0, 9.264491981582191, 5, 1, 6600.0, "1.1")]
    gold["gt_cases"] = np.array([[v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], float(v[8])] for v in gt_cases], np.float64)
    x = (rng2.standard_normal(700) * 2000).astype(np.float32)
    gold["gt_x"] = x
    dcf = dco = dfl = nfl = 0
    for i, v in enumerate(gt_cases):
        ch = v[5]
        for c in R:
            cf, coef, fl = np.zeros(ch, np.float32), np.zeros(4 * ch, np.float32), np.zeros(len(x) * ch, np.float32)
            assert R[c].ref_gammatone(v[0], v[1], v[2], v[3], v[4], ch, v[6], v[7], v[8].encode(), x, len(x), 256, cf, coef, fl) == 0
            gold["gt_cf_%d_%s" % (i, c)], gold["gt_coef_%d_%s" % (i, c)], gold["gt_out_%d_%s" % (i, c)] = cf, coef, fl
        dcf += ndiff(gold["gt_cf_%d_off" % i], gold["gt_cf_%d_fma" % i])
        dco += ndiff(gold["gt_coef_%d_off" % i], gold["gt_coef_%d_fma" % i])
        dfl += ndiff(gold["gt_out_%d_off" % i], gold["gt_out_%d_fma" % i])
        nfl += len(x) * ch
    report["Signal::GammaTone design: centre frequencies + coefficients (function text, GammaTone.cc:20-223)"] = dict(
        tried=5 * sum(v[5] for v in gt_cases), differ=dcf + dco,
        fma_sites="vfmsub132ss (WarpingFunction::init: factor * break - max), vfmadd (warping: beta * f + b), vfmadd132ss (xMin + i * scale)")
    report["Signal::GammaTone::apply, the cascade (the same pin)"] = dict(
        tried=nfl, differ=dfl, fma_sites="2 x vfnmadd132ss (out -= b1 * buffer0, out -= b2 * buffer1), vfmadd132ss (the second product of out * a0 + a1 * buffer0)")

    # ---- f4: the integration nodes' windows and arithmetic (function text: WindowFunction.cc + TemporalIntegration.cc + SpectralIntegration.cc)
    for c in R:
        R[c].ref_window_table.restype = C.c_int
        R[c].ref_window_table.argtypes = [C.c_int, C.c_int, f32p]
        R[c].ref_temporal_integration.restype = None
        R[c].ref_temporal_integration.argtypes = [C.c_int, f32p, C.c_int, C.c_int, f32p]
        R[c].ref_spectral_integration.restype = C.c_int
        R[c].ref_spectral_integration.argtypes = [C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int, f32p]
    d = 0
    for n in range(2, 2049):
        a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
        R["off"].ref_window_table(0, n, a)
        R["fma"].ref_window_table(0, n, b)
        d += ndiff(a, b)
        if n in (2, 3, 9, 160, 400):
            gold["hanning_%d" % n] = a
    report["Signal::HanningWindowFunction::init (function text, WindowFunction.cc:106-120), every length 2 .. 2048"] = dict(
        tried=sum(range(2, 2049)), differ=d, fma_sites="vfnmadd132sd (0.5 - 0.5 * cos()), f64; the f32 table hides it")
    ti_cases = [(0, 400, 12), (2, 37, 5), (0, 2, 3), (0, 161, 68)]   # (window: 0 Hanning / 2 rectangular, rows, channels)
    gold["ti_cases"] = np.array(ti_cases, np.int32)
    dt = nt = 0
    for i, (win, rows, ch) in enumerate(ti_cases):
        fr = (rng2.standard_normal((rows, ch)) * 500).astype(np.float32)
        gold["ti_in_%d" % i] = fr
        for c in R:
            out = np.zeros(ch, np.float32)
            R[c].ref_temporal_integration(win, fr.reshape(-1), rows, ch, out)
            gold["ti_out_%d_%s" % (i, c)] = out
        dt += ndiff(gold["ti_out_%d_off" % i], gold["ti_out_%d_fma" % i])
        nt += ch
    report["Signal::TemporalIntegration::transform (function text, TemporalIntegration.cc:22-81)"] = dict(
        tried=nt, differ=dt, fma_sites="vfmadd132sd (out += fabs(x) * w[i]: fabs is the double overload, product and sum in f64)")
    si_cases = [(0, 9, 4, 68), (2, 3, 1, 12), (0, 5, 5, 50)]   # (window, length, shift, channels)
    gold["si_cases"] = np.array(si_cases, np.int32)
    dsp = nsp = 0
    for i, (win, length, shift, ch) in enumerate(si_cases):
        x = (rng2.standard_normal((6, ch)) * 100).astype(np.float32)
        gold["si_in_%d" % i] = x
        oc = (ch - length) // shift + 1
        for c in R:
            out = np.zeros((6, oc), np.float32)
            assert R[c].ref_spectral_integration(win, length, shift, x.reshape(-1), 6, ch, out.reshape(-1)) == oc
            gold["si_out_%d_%s" % (i, c)] = out
        dsp += ndiff(gold["si_out_%d_off" % i], gold["si_out_%d_fma" % i])
        nsp += 6 * oc
    report["Signal::SpectralIntegration::apply (function text, SpectralIntegration.cc:25-75)"] = dict(
        tried=nsp, differ=dsp, fma_sites="vfmadd132ss (out += w[k] * in[ch * shift + k])")

    # ---- f1: Signal::Normalization + five algorithms on the reference's own sliding window (function text)
    for c in R:
        R[c].ref_normalization.restype = C.c_long
        R[c].ref_normalization.argtypes = [C.c_int, C.c_int, C.c_ulong, C.c_ulong, f32p, C.c_long, C.c_int, f32p]
    BIG = 2 ** 31 - 1
    ncases, dn, nn_ = [], 0, 0
    for trial in range(40):
        typ, n, dim = trial % 5, int(rng2.integers(1, 50)), int(rng2.integers(1, 14))
        level = int(rng2.integers(0, dim))
        if trial % 3 == 0:
            length, right = BIG, BIG
        else:
            length = int(rng2.integers(1, 20))
            right = int(rng2.integers(0, length))
        x = (rng2.standard_normal((n, dim)) * 10 + (5 if typ == 4 else 0)).astype(np.float32)
        ncases.append([typ, level, length, right, n, dim])
        gold["norm_in_%d" % trial] = x
        for c in R:
            out = np.zeros((n, dim), np.float32)
            assert R[c].ref_normalization(typ, level, length, right, x.reshape(-1), n, dim, out.reshape(-1)) == n
            gold["norm_out_%d_%s" % (trial, c)] = out
        dn += ndiff(gold["norm_out_%d_off" % trial], gold["norm_out_%d_fma" % trial])
        nn_ += n * dim
    gold["norm_cases"] = np.array(ncases, np.int64)
    report["Signal::Normalization: level / mean / mean-and-variance / -1D / divide-by-mean on Signal::SlidingWindow (function text, "
           "Normalization.cc:24-262 + SlidingWindow.hh:22-471), 40 segments"] = dict(
        tried=nn_, differ=dn, fma_sites="f64 products of widened f32 values are exact: fused or not, the same bits")

    # ---- f4: Mm::DensityClustering<f32, f32> (function text): rand()-initialised k-means over the scaled means, cluster selection
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
    for c in R:
        R[c].ref_density_clustering.restype = C.c_int
        R[c].ref_density_clustering.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, f32p, f32p, C.c_int, u8p]
    dc_cases = [(90, 24, 16, 4), (300, 40, 64, 10), (20, 40, 256, 32), (64, 8, 8, 8)]   # (densities, padded dim, clusters, select)
    gold["dc_cases"] = np.array(dc_cases, np.int32)
    dd = nd = 0
    for i, (nk, pdim, ncl, nsel) in enumerate(dc_cases):
        ms = (rng2.standard_normal((nk, pdim)) * 3).astype(np.float32)
        xs = (rng2.standard_normal((12, pdim)) * 3).astype(np.float32)
        gold["dc_ms_%d" % i], gold["dc_xs_%d" % i] = ms, xs
        nce = min(ncl, nk)
        for c in R:
            cof, cm, sel = np.zeros(nk, np.uint8), np.zeros(nce * pdim, np.float32), np.zeros(12 * nce, np.uint8)
            assert R[c].ref_density_clustering(ms.reshape(-1), nk, pdim, ncl, min(nsel, nce), 5, cof, cm, xs.reshape(-1), 12, sel) == nce
            gold["dc_cof_%d_%s" % (i, c)], gold["dc_cm_%d_%s" % (i, c)], gold["dc_sel_%d_%s" % (i, c)] = cof, cm, sel
        dd += ndiff(gold["dc_cm_%d_off" % i], gold["dc_cm_%d_fma" % i]) + int(np.count_nonzero(gold["dc_cof_%d_off" % i] != gold["dc_cof_%d_fma" % i]))
        nd += nce * pdim + nk
    R["off"].ref_density_clustering_u8.restype = C.c_int
    R["off"].ref_density_clustering_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p]
    for i, (nk, dim, ncl) in enumerate([(150, 48, 32), (40, 16, 256), (300, 64, 100)]):
        q = rng2.integers(0, 256, (nk, dim)).astype(np.uint8)
        if i == 0:
            q[nk // 2:] = q[:nk - nk // 2]   # duplicates: ties
        nce = min(ncl, nk)
        cof, cm = np.zeros(nk, np.uint8), np.zeros(nce * dim, np.uint8)
        assert R["off"].ref_density_clustering_u8(q.reshape(-1), nk, dim, ncl, 5, cof, cm) == nce
        gold["dcu_ms_%d" % i], gold["dcu_cof_%d" % i], gold["dcu_cm_%d" % i], gold["dcu_ncl_%d" % i] = q, cof, cm, np.array([ncl])
    report["Mm::DensityClustering<f32, f32> (function text, DensityClustering.tcc:61-119,157-180 + DensityClustering.cc:45-57): assignment + means"] = dict(
        tried=nd, differ=dd, fma_sites="unrolledVectorDistance's score += df * df (the product has no contract=fma mode for the preselection scorers)")

    # ---- a13 / a14: calculateScoreAndDensity of the maximum and of the log-add scorer (function text), per (frame, mixture), fed with
    # the f32 tables the oracle builds from the model (those tables have their own pins: gaussLogNormFactor, inverseSquareRoot)
    from oracle import OracleGmm
    sys.path.insert(0, ROOT)
    from tests import synth
    for c in R:
        R[c].ref_gdm_score.restype = None
        R[c].ref_gdm_score.argtypes = [C.c_int, f32p, C.c_int, C.c_int, f32p, f32p, f32p, f32p, C.POINTER(C.c_float), C.POINTER(C.c_uint), C.c_void_p]
    sc_cases = [(40, 16, True, 905), (39, 7, False, 906), (16, 3, True, 907), (45, 12, False, 908)]   # (dim, max densities, pooled, seed)
    gold["sc_cases"] = np.array([[a, b, int(p_), sd] for a, b, p_, sd in sc_cases], np.int32)
    dsc = nsc = 0
    for i, (dim, kmax, pooled, seed) in enumerate(sc_cases):
        model = synth.gmm_cart(6, 1, kmax, dim, seed=seed, pooled=pooled)
        off = model["mix_offsets"]
        if off[1] - off[0] >= 2:   # a tie inside mixture 0: the same density listed twice with the same weight
            model["dens_index"][off[0] + 1] = model["dens_index"][off[0]]
            model["log_weight"][off[0] + 1] = model["log_weight"][off[0]]
        x = rng2.standard_normal((8, dim)).astype(np.float32)
        x[6, 0] = np.nan
        x[7] *= 1e19
        for k, v in model.items():
            if isinstance(v, np.ndarray):
                gold["sc_model_%d_%s" % (i, k)] = v
        gold["sc_x_%d" % i] = x
        for c in R:
            g = OracleGmm(model, contract=c)
            m2lw, isr, ln = g.tables()
            for mode in (0, 1):
                sc, best = np.zeros((8, 6), np.float32), np.zeros((8, 6), np.uint32)
                for t in range(8):
                    for m in range(6):
                        k0, k1 = int(off[m]), int(off[m + 1])
                        d = model["dens_index"][k0:k1]
                        means = np.ascontiguousarray(model["means"][model["dens_mean"][d]], np.float32)
                        isrs = np.ascontiguousarray(isr[model["dens_cov"][d]], np.float32)
                        lns = np.ascontiguousarray(ln[model["dens_cov"][d]], np.float32)
                        s_, b_ = C.c_float(0), C.c_uint(0)
                        R[c].ref_gdm_score(mode, np.ascontiguousarray(x[t]), dim, k1 - k0, np.ascontiguousarray(m2lw[k0:k1]), lns,
                                           means.reshape(-1), isrs.reshape(-1), C.byref(s_), C.byref(b_), None)
                        sc[t, m], best[t, m] = s_.value, b_.value
                gold["sc_score_%d_%d_%s" % (i, mode, c)], gold["sc_best_%d_%d_%s" % (i, mode, c)] = sc, best
        for mode in (0, 1):
            dsc += ndiff(gold["sc_score_%d_%d_off" % (i, mode)], gold["sc_score_%d_%d_fma" % (i, mode)])
            nsc += 48
    report["Mm::GaussDiagonalMaximumFeatureScorer::calculateScoreAndDensity + GaussDiagonalSumFeatureScorer (function text, "
           "GaussDiagonalMaximumFeatureScorer.cc:116-142,230-298) on the distance of the same pin"] = dict(
        tried=nsc, differ=dsc, fma_sites="none of their own (f64 sums, one product by 0.5); the distance's")

    # ---- a9 and the rest of generic-vector-f32-<function> (templates of Flow/SimpleFunction.hh taken whole): 13 kinds incl. NaN / inf / +-0
    rng3 = np.random.default_rng(20261001)
    for c in R:
        R[c].ref_vector_function.restype = None
        R[c].ref_vector_function.argtypes = [C.c_int, C.c_float, f32p, C.c_long, f32p]
    dv = 0
    for kind in range(13):
        x = (rng3.standard_normal(96) * rng3.choice([0.01, 1, 100, 1e6], 96)).astype(np.float32)
        if kind in (0, 1, 2, 4, 5):
            x = np.abs(x) + np.float32(1e-3)
        x[:5] = [0.0, -0.0, np.inf, -np.inf, np.nan]
        prm = np.float32([0.0, 0.33, 0.0, 0.0, 0.33, 0.0, 0.0, 2.5, 0.1, 0.25, 0.0, 1.5, -1.5][kind])
        gold["vf_in_%d" % kind], gold["vf_prm_%d" % kind] = x, np.array([prm], np.float32)
        for c in R:
            out = np.zeros_like(x)
            R[c].ref_vector_function(kind, prm, x, len(x), out)
            gold["vf_out_%d_%s" % (kind, c)] = out
        a, b = gold["vf_out_%d_off" % kind], gold["vf_out_%d_fma" % kind]
        dv += int(np.count_nonzero((bits(a) != bits(b)) & ~(np.isnan(a) & np.isnan(b))))
    report["generic-vector-f32-<function>: 13 functors of Flow/SimpleFunction.hh (templates taken whole, :32-358)"] = dict(tried=13 * 96, differ=dv)

    # ---- f1: signal-vector-f32-<kind>-normalization (templates of Signal/VectorNormalization.hh taken whole)
    for c in R:
        R[c].ref_vector_normalize.restype = None
        R[c].ref_vector_normalize.argtypes = [C.c_int, f32p, C.c_int, f32p]
    xv = (rng3.standard_normal((40, 33)) * rng3.choice([0.01, 1, 1000], (40, 1))).astype(np.float32)
    gold["vn_in"] = xv
    dvn = 0
    for typ in range(6):
        for c in R:
            out = np.zeros_like(xv)
            for r in range(len(xv)):
                R[c].ref_vector_normalize(typ, xv[r], xv.shape[1], out[r])
            gold["vn_out_%d_%s" % (typ, c)] = out
        dvn += ndiff(gold["vn_out_%d_off" % typ], gold["vn_out_%d_fma" % typ])
    report["signal-vector-f32-<kind>-normalization: 6 functors of Signal/VectorNormalization.hh (templates taken whole, :35-171)"] = dict(
        tried=6 * xv.size, differ=dvn, fma_sites="vfmadd (amplitude-spectrum-energy: front * front + back * back)")

    np.savez_compressed(os.path.join(HERE, "ref_contract.npz"), **gold)
    out = os.path.join(ROOT, "profiles", "r05")
    os.makedirs(out, exist_ok=True)
    import subprocess
    meta = dict(host_march=subprocess.run("gcc -march=native -Q --help=target | grep -m1 march=", shell=True, capture_output=True, text=True).stdout.split()[-1],
                gcc=subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0],
                flavours=dict(off="-std=c++20 -O2 -msse3 (the reference with -DMARCH=x86-64)", fma="-std=c++20 -O2 -msse3 -march=native (the reference's default)"),
                note="differ = outputs whose bits differ between the two flavours of the compiled reference on the same inputs")
    json.dump(dict(meta=meta, pins=report), open(os.path.join(out, "contract_pins.json"), "w"), indent=1)
    for k, v in report.items():
        print("%6d / %-7d %s" % (v["differ"], v["tried"], k))


if __name__ == "__main__":
    main()
