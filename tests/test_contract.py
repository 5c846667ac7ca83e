"""CPU: the oracle's TWO builds against the reference's two builds.

The reference's default configuration (-march=native, GCC's -ffp-contract=fast) fuses `sum += df * df` of the GMM distance and a few
other f32 sums into fused multiply-adds; configured with -DMARCH=x86-64 it does not.  oracle/liboracle.so restates the second,
oracle/liboracle_fma.so the first (orc.h).  Both are held, bit for bit, to
  * tests/golden/ref_contract.npz -- outputs of the reference compiled both ways (tests/golden/make_contract_golden.py), everywhere;
  * oracle/_ref/libref.so / libref_native.so live on fresh inputs, where the reference tree is mounted (build container).
Function-text pins (oracle/ref/extract_fn.py): a13 / a14 distance, combine rule and log-add scorer, f1 regression, a7 filter apply, a3 Hamming table, a18 batch-float sum, a1 preemphasis, a7 filter builder and boundary, a10 cosine transform, f4 AR-to-cepstrum, the gammatone filter bank and the integration nodes, f1 normalisation, f4 density clustering.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import OracleGmm
from oracle.binding import CONTRACTS, Oracle, load_ref, oracle_matrix_multiply, oracle_regression
from tests import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
HAVE_REF = os.path.isdir("/root/reference/src")
Z = np.load(os.path.join(GOLD, "ref_contract.npz"))
DIMS = [40, 39, 33, 32, 24, 16, 7, 3, 45]


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64)


def test_the_two_libraries_are_the_two_builds():
    assert Oracle("off").orc_contract() == 0 and Oracle("fma").orc_contract() == 1
    assert Oracle("off") is not Oracle("fma")


@pytest.mark.parametrize("contract", CONTRACTS)
def test_gmm_distance_against_the_reference_function_text(contract):
    """a13 / a14: Mm::GaussDiagonalMaximumFeatureScorer::distance -- the reference's own function text compiled with that build's flags"""
    L = Oracle(contract)
    for dim in DIMS:
        x, mu, isr = Z["dist_x_%d" % dim], Z["dist_mu_%d" % dim], Z["dist_isr_%d" % dim]
        got = np.array([L.orc_gmm_distance(x[i], mu[i], isr[i], dim) for i in range(len(x))], np.float32)
        assert np.array_equal(bits(got), bits(Z["dist_%d_%s" % (dim, contract)])), (contract, dim)


def test_the_two_builds_of_the_distance_really_differ():
    """about a fifth of the d = 40 distances differ in bits between the builds, by one unit in the last place or so"""
    a, b = Z["dist_40_off"], Z["dist_40_fma"]
    frac = np.count_nonzero(bits(a) != bits(b)) / a.size
    assert 0.1 < frac < 0.4, frac
    assert np.max(np.abs(a.astype(np.float64) - b) / np.abs(a)) < 5e-7


@pytest.mark.parametrize("contract", CONTRACTS)
def test_regression_against_the_reference_function_text(contract):
    """f1: Signal::Regression::regressFirstOrder / regressSecondOrder (first pin of the back end's regression)"""
    for order in (1, 2):
        for right in (1, 2, 3):
            w = Z["reg_in_%d_%d" % (order, right)]
            want = Z["reg_%d_%d_%s" % (order, right, contract)]
            for i in range(len(w)):
                got = oracle_regression(w[i], order=order, right=right, contract=contract)[right]   # the middle frame sees the whole window
                assert np.array_equal(bits(got), bits(want[i])), (contract, order, right, i)


@pytest.mark.parametrize("contract", CONTRACTS)
def test_filter_bank_apply_against_the_reference_function_text(contract):
    """a7: Signal::FilterBank::Filter::apply"""
    L = Oracle(contract)
    got = np.array([L.orc_filter_apply(Z["fb_amp"][i], int(Z["fb_start"][i]), int(Z["fb_end"][i]), Z["fb_w"][i]) for i in range(200)], np.float32)
    assert np.array_equal(bits(got), bits(Z["fb_%s" % contract]))


@pytest.mark.parametrize("contract", CONTRACTS)
def test_hamming_table_against_the_reference_function_text(contract):
    """a3: Signal::HammingWindowFunction::init.  The default build fuses 0.54 - 0.46 * cos() in f64; the f32 table has the same bits in
    both builds for every length 2 .. 4096 (contract_pins.json), so one restatement serves both"""
    L = Oracle(contract)
    for n in (2, 3, 160, 200, 320, 400, 401, 512, 1001):
        w = np.zeros(n, np.float32)
        L.orc_hamming_window(w, n)
        assert np.array_equal(bits(w), bits(Z["hamming_%d" % n])), (contract, n)


@pytest.mark.parametrize("contract", CONTRACTS)
def test_preemphasis_against_the_reference_function_text(contract):
    """a1: Signal::Preemphasis, the whole class, fed 4096-sample blocks with contiguous time stamps (the state carries over) and one
    restart behind a gap (previous = the block's first sample); alpha = 1 is a plain difference, alpha != 1 has the one product that the
    default build fuses"""
    L = Oracle(contract)
    x = Z["pre_x"]
    for name, alpha in (("1", 1.0), ("097", 0.97), ("05", 0.5)):
        y = x.copy()
        L.orc_preemphasis(y, len(y), alpha)
        assert np.array_equal(bits(y), bits(Z["pre_%s_%s" % (name, contract)])), (contract, name)
    a, b = x[:4096].copy(), x[4096:].copy()
    L.orc_preemphasis(a, len(a), 0.97)
    L.orc_preemphasis(b, len(b), 0.97)
    assert np.array_equal(bits(np.concatenate([a, b])), bits(Z["pre_097_gap_%s" % contract]))
    assert np.array_equal(bits(Z["pre_1_off"]), bits(Z["pre_1_fma"])) and not np.array_equal(bits(Z["pre_097_off"]), bits(Z["pre_097_fma"]))


def _filter_build(fn, p):
    st, en, w = C.c_int(0), C.c_int(0), np.zeros(600, np.float32)
    n = fn(int(p[0]), int(p[1]), float(p[2]), float(p[3]), float(p[4]), float(p[5]), float(p[6]), int(p[7]), C.byref(st), C.byref(en), w, 600)
    return n, (st.value if n >= 0 else -1), (en.value if n >= 0 else -1), w[:max(n, 0)].copy()


@pytest.mark.parametrize("contract", CONTRACTS)
def test_filter_builder_against_the_reference_function_text(contract):
    """a7: FilterBank::FilterBuilder::create for ONE filter -- where it starts (ceil, or round on an almost-integer bin), where it ends,
    its weights (f32 shape x f64 derivative of the warping), and when the builder refuses -- triangular / trapeze, mel / bark, with and
    without the differential unit; the analytic functions are the reference's own classes"""
    L = Oracle(contract)
    P = Z["fbb_params"]
    n_w, off, refused = Z["fbb_n_%s" % contract], 0, 0
    for i in range(len(P)):
        n, st, en, w = _filter_build(L.orc_filter_build, P[i])
        assert (n < 0) == (n_w[i] < 0), (contract, i)
        if n < 0:
            refused += 1
            continue
        assert n == n_w[i] and st == Z["fbb_start_%s" % contract][i] and en == Z["fbb_end_%s" % contract][i], (contract, i, P[i])
        assert np.array_equal(bits(w), bits(Z["fbb_w_%s" % contract][off:off + n])), (contract, i, P[i])
        off += n
    assert refused > 10 and off == len(Z["fbb_w_%s" % contract])


def _boundary(fn, p):
    w, sp, cen = C.c_double(0), C.c_double(0), np.zeros(256)
    n = fn(int(p[0]), float(p[1]), float(p[2]), float(p[3]), float(p[4]), float(p[5]), C.byref(w), C.byref(sp), cen, 256)
    return n, w.value, sp.value, cen[:max(0, min(n, 256))].copy()


@pytest.mark.parametrize("contract", CONTRACTS)
def test_filter_bank_boundary_against_the_reference_function_text(contract):
    """a7: how many filters a bank has, the width and spacing the boundary ends up with (stretch-to-cover divides both by the coverage) and
    the centres, for the three boundary types; f64, and NOT the same in the two builds: the default build fuses spacing * (n - 1) + width
    and the products of the stretch-to-cover centre"""
    L = Oracle(contract)
    P, off = Z["fbd_params"], 0
    for i in range(len(P)):
        n, w, sp, cen = _boundary(L.orc_filter_boundary, P[i])
        assert n == Z["fbd_n_%s" % contract][i], (contract, i, P[i])
        assert np.array_equal(bits(np.array([w, sp])), bits(Z["fbd_ws_%s" % contract][i])), (contract, i, P[i])
        k = min(n, 256)
        assert np.array_equal(bits(cen), bits(Z["fbd_centers_%s" % contract][off:off + k])), (contract, i, P[i])
        off += k
    assert off == len(Z["fbd_centers_%s" % contract])
    assert not np.array_equal(bits(Z["fbd_centers_off"]), bits(Z["fbd_centers_fma"])) and np.array_equal(Z["fbd_n_off"], Z["fbd_n_fma"])


@pytest.mark.parametrize("contract", CONTRACTS)
def test_cosine_transform_against_the_reference_function_text(contract):
    """a10: Signal::CosineTransform, the whole class -- both tables (even about N - 1/2: the MFCC's DCT; N plus one: the autocorrelation of
    MF-PLP / PLP) and apply, with and without the division by N"""
    L = Oracle(contract)
    for i, (np1, n_in, n_out, norm) in enumerate(Z["ct_cases"]):
        x = Z["ct_in_%d" % i]
        tab = np.zeros(int(n_out) * int(n_in), np.float32)
        for r in range(len(x)):
            out = np.zeros(int(n_out), np.float32)
            L.orc_cosine_transform(int(np1), int(n_in), int(n_out), int(norm), np.ascontiguousarray(x[r]), out, tab)
            assert np.array_equal(bits(out), bits(Z["ct_out_%d_%s" % (i, contract)][r])), (contract, i, r)
        assert np.array_equal(bits(tab), bits(Z["ct_tab_%d_%s" % (i, contract)])), (contract, i)
        assert np.array_equal(bits(Z["ct_tab_%d_off" % i]), bits(Z["ct_tab_%d_fma" % i]))   # the tables do not depend on the build


@pytest.mark.parametrize("contract", CONTRACTS)
def test_autoregression_to_cepstrum_against_the_reference_function_text(contract):
    """f4 (MF-PLP / PLP): Signal::autoregressionToCepstrum; the default build fuses the second product of c[n] += (n - k) c[n - k] a[k - 1]"""
    from oracle.binding import oracle_ar_to_cepstrum
    A, G = Z["arc_a"], Z["arc_gain"]
    for i in range(len(A)):
        got = oracle_ar_to_cepstrum(float(G[i]), A[i], 16, contract=contract)
        assert np.array_equal(bits(got), bits(Z["arc_%s" % contract][i])), (contract, i)
    assert not np.array_equal(bits(Z["arc_off"]), bits(Z["arc_fma"]))


@pytest.mark.parametrize("contract", CONTRACTS)
def test_gammatone_filter_bank_against_the_reference_function_text(contract):
    """f4: signal-gammatone -- Signal::WarpingFunction and Signal::GammaTone, design (centre frequencies on the Greenwood / ERB scale
    through the two-piece warping, bandwidths, coefficients) and the cascade with its state over blocks; the default build fuses three
    operations of the design and three of the four products of a cascade stage"""
    from oracle.binding import GammatoneCfg, OracleGammatone
    x = Z["gt_x"]
    for i, v in enumerate(Z["gt_cases"]):
        cfg = GammatoneCfg.default(sample_rate=float(v[0]), cascade=int(v[1]), min_freq=float(v[2]), max_freq=float(v[3]), q=float(v[4]),
                                   channels=int(v[5]), cf_mode=int(v[6]), warp_freq_break=float(v[7]), warping_factor=float(v[8]))
        o = OracleGammatone(cfg, contract=contract)
        _, filt = o.run(x, want_filtered=True)
        assert np.array_equal(bits(o.center_frequencies), bits(Z["gt_cf_%d_%s" % (i, contract)])), (contract, i)
        assert np.array_equal(bits(o.coefficients.reshape(-1)), bits(Z["gt_coef_%d_%s" % (i, contract)])), (contract, i)
        assert np.array_equal(bits(filt.reshape(-1)), bits(Z["gt_out_%d_%s" % (i, contract)])), (contract, i)
    assert not np.array_equal(bits(Z["gt_out_0_off"]), bits(Z["gt_out_0_fma"]))


@pytest.mark.parametrize("contract", CONTRACTS)
def test_integration_nodes_against_the_reference_function_text(contract):
    """f4: the Hanning / rectangular tables of Signal/WindowFunction.cc, Signal::TemporalIntegration::transform (|x| weighted by the window,
    summed in f64 because fabs is the double overload there) and Signal::SpectralIntegration::apply (f32)"""
    L = Oracle(contract)
    for n in (2, 3, 9, 160, 400):
        w = np.array([L.orc_window_value(0, n, i) for i in range(n)], np.float32)
        assert np.array_equal(bits(w), bits(Z["hanning_%d" % n])), n
    for i, (win, rows, ch) in enumerate(Z["ti_cases"]):
        out = np.zeros(int(ch), np.float32)
        L.orc_temporal_integrate(0 if win == 0 else 1, np.ascontiguousarray(Z["ti_in_%d" % i]).reshape(-1), int(rows), int(ch), out)
        assert np.array_equal(bits(out), bits(Z["ti_out_%d_%s" % (i, contract)])), (contract, i)
    for i, (win, length, shift, ch) in enumerate(Z["si_cases"]):
        wt = np.array([L.orc_window_value(0 if win == 0 else 1, int(length), k) for k in range(int(length))], np.float32)
        x, want = Z["si_in_%d" % i], Z["si_out_%d_%s" % (i, contract)]
        for r in range(len(x)):
            out = np.zeros(want.shape[1], np.float32)
            assert L.orc_spectral_integrate(wt, int(length), int(shift), np.ascontiguousarray(x[r]), int(ch), out) == want.shape[1]
            assert np.array_equal(bits(out), bits(want[r])), (contract, i, r)
    # (the temporal sum's fused f64 operation hides behind the narrowing to f32 at every step: the builds agree on the fixture's 88 values;
    # the spectral sum is f32 and differs)
    assert not np.array_equal(bits(Z["si_out_0_off"]), bits(Z["si_out_0_fma"]))


@pytest.mark.parametrize("contract", CONTRACTS)
def test_normalization_against_the_reference_function_text(contract):
    """f1: Signal::Normalization with level / mean / mean-and-variance / mean-and-variance-1D / divide-by-mean on the reference's own
    Signal::SlidingWindow, fed and flushed as NormalizationNode::work does: the arithmetic AND which frame leaves when, with which
    statistics (finite windows with a look-ahead, and the whole segment)"""
    from oracle.binding import oracle_normalize_ex
    omap = {0: 3, 1: 0, 2: 1, 3: 4, 4: 2}   # the reference's type order -> orc_normalize_ex's
    for i, (typ, level, length, right, n, dim) in enumerate(Z["norm_cases"]):
        whole = length >= 2 ** 31 - 1
        got = oracle_normalize_ex(Z["norm_in_%d" % i], omap[int(typ)], level=int(level), length=0 if whole else int(length),
                                  right=0 if whole else int(right), contract=contract)
        want = Z["norm_out_%d_%s" % (i, contract)]
        assert _same_bits_or_both_nan(got, want), (contract, i, typ, length, right)


@pytest.mark.parametrize("contract", CONTRACTS)
def test_score_and_best_density_against_the_reference_function_text(contract):
    """a13 / a14: calculateScoreAndDensity of Mm::GaussDiagonalMaximumFeatureScorer (f64 sum of the three terms, `bestScore > score`
    against the narrowed best, 0.5 x, first density on ties, 0.5 x FLT_MAX and "no density" on NaN or overflowing frames) and of the log-add scorer (f32 sum, expf / logf):
    the reference's own function text, fed per (frame, mixture) with the oracle's f32 tables"""
    for i, (dim, kmax, pooled, seed) in enumerate(Z["sc_cases"]):
        model = {k[len("sc_model_%d_" % i):]: Z[k] for k in Z.files if k.startswith("sc_model_%d_" % i)}
        model["dim"] = int(dim)
        g = OracleGmm(model, contract=contract)
        for mode in (0, 1):
            sc, best = g.score(Z["sc_x_%d" % i], mode=mode)
            want, wbest = Z["sc_score_%d_%d_%s" % (i, mode, contract)], Z["sc_best_%d_%d_%s" % (i, mode, contract)]
            assert _same_bits_or_both_nan(sc, want), (contract, i, mode)
            assert np.array_equal(best.astype(np.uint32), wbest), (contract, i, mode)
        # a NaN frame (row 6) and one whose distances overflow (row 7): `bestScore > score` is never true -- 0.5 x FLT_MAX and no density
        for row in (6, 7):
            assert (Z["sc_score_%d_0_%s" % (i, contract)][row] == np.float32(0.5 * float(np.finfo(np.float32).max))).all()
            assert (Z["sc_best_%d_0_%s" % (i, contract)][row] == 0xffffffff).all()


def test_density_clustering_against_the_reference_function_text():
    """f4 (preselection-batch-float): Mm::DensityClustering<f32, f32> -- srand(1) / rand() initialisation, five k-means iterations
    (assignment by unrolledVectorDistance, means through f64 sums), the cluster count reduced to the number of densities, and
    selectClusters.  x86-64 build only: the preselection scorers have no contract=fma mode (the default build fuses the distance's sum)"""
    L = Oracle("off")
    for i, (nk, pdim, ncl, nsel) in enumerate(Z["dc_cases"]):
        ms, xs = Z["dc_ms_%d" % i], Z["dc_xs_%d" % i]
        nk, pdim = int(nk), int(pdim)
        # a model whose scaled, padded means ARE ms: unit pooled variance (1 / sqrt(1) = 1 exactly), one density per mixture
        model = dict(dim=pdim, mix_offsets=np.arange(nk + 1, dtype=np.uint32), dens_index=np.arange(nk, dtype=np.uint32),
                     log_weight=np.zeros(nk), dens_mean=np.arange(nk, dtype=np.uint32), dens_cov=np.zeros(nk, np.uint32),
                     means=np.ascontiguousarray(ms), variances=np.ones((1, pdim), np.float32))
        nce = min(int(ncl), nk)
        _, cof, cm = OracleGmm(model, contract="off").score_preselection_float(xs, int(ncl), min(int(nsel), nce), 5, 40000.0)
        assert np.array_equal(np.asarray(cof).astype(np.uint8), Z["dc_cof_%d_off" % i]), i
        assert np.array_equal(bits(np.ascontiguousarray(cm, np.float32).reshape(-1)), bits(Z["dc_cm_%d_off" % i])), i
        want = Z["dc_sel_%d_off" % i].reshape(len(xs), nce)
        for t in range(len(xs)):
            sel = np.zeros(nce, np.uint8)
            L.orc_cluster_select(np.ascontiguousarray(Z["dc_cm_%d_off" % i]), nce, pdim, min(int(nsel), nce), np.ascontiguousarray(xs[t]), sel)
            assert np.array_equal(sel, want[t]), (i, t)


@pytest.mark.parametrize("contract", CONTRACTS)
def test_vector_functions_against_the_reference_templates(contract):
    """a9 (the MFCC's log10) and the other generic-vector-f32-<function> functors: Flow/SimpleFunction.hh's templates taken whole; which
    libm overload the unqualified log10 / pow / rint pick on an f32, min / max with NaN, +-0 and infinities"""
    L = Oracle(contract)
    L.orc_vector_function.restype = None
    L.orc_vector_function.argtypes = [C.c_int, C.c_float, np.ctypeslib.ndpointer(np.float32, flags="C"), C.c_long, C.c_int,
                                      np.ctypeslib.ndpointer(np.float32, flags="C")]
    for kind in range(13):
        x, prm = Z["vf_in_%d" % kind], float(Z["vf_prm_%d" % kind][0])
        out = np.zeros_like(x)
        L.orc_vector_function(kind, prm, np.ascontiguousarray(x), len(x), 1, out)
        assert _same_bits_or_both_nan(out, Z["vf_out_%d_%s" % (kind, contract)]), (contract, kind)


@pytest.mark.parametrize("contract", CONTRACTS)
def test_vector_normalization_against_the_reference_templates(contract):
    """f1: signal-vector-f32-<kind>-normalization -- the six functors of Signal/VectorNormalization.hh taken whole (sums with a double seed
    over f32-rounded products, statistics narrowed to f32, f32 scaling); the default build fuses one product of the amplitude-spectrum
    energy's end terms"""
    L = Oracle(contract)
    L.orc_vector_normalize.restype = None
    L.orc_vector_normalize.argtypes = [C.c_int, np.ctypeslib.ndpointer(np.float32, flags="C"), C.c_int, C.c_int,
                                       np.ctypeslib.ndpointer(np.float32, flags="C")]
    x = Z["vn_in"]
    for typ in range(6):
        out = np.zeros_like(x)
        L.orc_vector_normalize(typ, np.ascontiguousarray(x).reshape(-1), x.shape[0], x.shape[1], out.reshape(-1))
        assert np.array_equal(bits(out), bits(Z["vn_out_%d_%s" % (typ, contract)])), (contract, typ)


def test_integer_density_clustering_against_the_reference_function_text():
    """f4 (preselection-batch-int): the same template as Mm::DensityClustering<u8, s32> -- integer distances, first cluster on ties (the
    first case repeats half of its entries), means through f64 sums converted to u8"""
    L = Oracle("off")
    for i in range(3):
        q, ncl = Z["dcu_ms_%d" % i], int(Z["dcu_ncl_%d" % i][0])
        nk, dim = q.shape
        nce = min(ncl, nk)
        cof, cm = np.zeros(nk, np.uint32), np.zeros(nce * dim, np.uint8)
        L.orc_cluster_u8(np.ascontiguousarray(q).reshape(-1), nk, dim, nce, 5, cof, cm)
        assert np.array_equal(cof.astype(np.uint8), Z["dcu_cof_%d" % i]) and np.array_equal(cm, Z["dcu_cm_%d" % i]), i


def _same_bits_or_both_nan(a, b):
    return bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("contract", CONTRACTS)
def test_batch_float_fill_against_the_reference_function_text(contract):
    """a18: Mm::BatchFloatFeatureScorer::fillScoreCacheTpl -- the SSE accumulate (two vfmadd in the default build), the horizontal sum, the
    minimum as `_mm_min_ps(score, s)` (a NaN sum REPLACES the score, a later finite or infinite one replaces the NaN) and the final 0.5"""
    L = Oracle(contract)
    n_nonfinite = 0
    for i in range(int(Z["bf_n"][0])):
        ms, cst, xs = Z["bf_ms_%d" % i], Z["bf_cst_%d" % i], Z["bf_xs_%d" % i]
        got = np.array([L.orc_batch_float_fill(ms.reshape(-1), cst, len(ms), np.ascontiguousarray(xs[t]), ms.shape[1]) for t in range(len(xs))],
                       np.float32)
        want = Z["bf_%d_%s" % (i, contract)]
        assert _same_bits_or_both_nan(got, want), (contract, i, got, want)
        n_nonfinite += int(np.count_nonzero(~np.isfinite(want)))
    assert n_nonfinite >= 10   # the cases that tell min_ps(score, s) from min(s, score) are in the fixture


def test_the_two_builds_of_the_batch_float_sum_really_differ():
    d = sum(int(np.count_nonzero(bits(Z["bf_%d_off" % i]) != bits(Z["bf_%d_fma" % i]))) for i in range(int(Z["bf_n"][0])))
    assert d > 20, d


@pytest.mark.parametrize("contract", CONTRACTS)
def test_matrix_times_vector_against_the_reference(contract):
    """a10 (cosine transform) / f1 (signal-matrix-multiplication): Math::Matrix<f32> * Math::Vector<f32>"""
    got = oracle_matrix_multiply(Z["mv_M"], Z["mv_v"], contract=contract)
    assert np.array_equal(bits(got), bits(Z["mv_%s" % contract]))


@pytest.mark.parametrize("contract", CONTRACTS)
def test_log_norm_factor_against_the_reference(contract):
    """a15: Mm::gaussLogNormFactor narrowed to f32 as CovarianceFeatureScorerElement stores it"""
    for dim in (40, 39, 33, 24, 13):
        var = Z["ln_var_%d" % dim]
        m = synth.gmm_cart(4, 1, 1, dim, seed=3, pooled=False)
        for i in range(0, len(var), 4):
            m["variances"] = np.ascontiguousarray(var[i:i + 4])
            m["dens_cov"] = np.arange(4, dtype=np.uint32)
            _, _, ln = OracleGmm(m, contract=contract).tables()
            assert np.array_equal(bits(ln), bits(Z["ln_%d_%s" % (dim, contract)][i:i + 4].astype(np.float32))), (contract, dim, i)


@pytest.mark.parametrize("contract", CONTRACTS)
def test_scores_follow_the_distance_of_their_build(contract):
    """the scorer of each build = its own distance + the (contraction-free) f64 combine: recomputed here from orc_gmm_distance"""
    m = synth.gmm_cart(12, 1, 6, 40, seed=5, pooled=True)
    x = np.random.default_rng(1).standard_normal((16, 40)).astype(np.float32)
    g = OracleGmm(m, contract=contract)
    sc, best = g.score(x)
    m2lw, isr, ln = g.tables()
    L = Oracle(contract)
    for t in range(16):
        for mix in range(12):
            k0, k1 = int(m["mix_offsets"][mix]), int(m["mix_offsets"][mix + 1])
            b, bi = np.float32(np.finfo(np.float32).max), 0xffffffff
            for k in range(k0, k1):
                d = int(m["dens_index"][k])
                dist = np.float32(L.orc_gmm_distance(x[t], np.ascontiguousarray(m["means"][m["dens_mean"][d]]), np.ascontiguousarray(isr[m["dens_cov"][d]]), 40))
                s = float(m2lw[k]) + float(ln[m["dens_cov"][d]]) + float(dist)
                if float(b) > s:
                    b, bi = np.float32(s), k - k0
            assert bits(np.float32(0.5 * float(b))) == bits(sc[t, mix]) and bi == best[t, mix]


def test_scores_of_the_two_builds_differ_by_rounding_only():
    m = synth.gmm_cart(64, 4, 16, 40, seed=2, pooled=True)
    x = np.random.default_rng(2).standard_normal((64, 40)).astype(np.float32)
    a, ba = OracleGmm(m, contract="off").score(x)
    b, bb = OracleGmm(m, contract="fma").score(x)
    assert np.count_nonzero(bits(a) != bits(b)) > 0
    assert np.max(np.abs(a.astype(np.float64) - b) / np.abs(a)) < 1e-6     # 1e-4 (north_star's bar) holds across builds by two orders
    assert np.array_equal(a.argmin(axis=1), b.argmin(axis=1))


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not mounted")
@pytest.mark.parametrize("contract", CONTRACTS)
def test_live_against_the_compiled_reference(contract):
    """fresh random inputs through the reference compiled with this build's flags (oracle/_ref)"""
    R, L = load_ref(contract), Oracle(contract)
    assert R is not None and hasattr(R, "ref_gdm_distance")
    rng = np.random.default_rng(77)
    for dim in (40, 39, 33, 5):
        for _ in range(300):
            x, mu = rng.standard_normal(dim).astype(np.float32), rng.standard_normal(dim).astype(np.float32)
            isr = rng.uniform(0.3, 3, dim).astype(np.float32)
            assert bits(np.float32(L.orc_gmm_distance(x, mu, isr, dim))) == bits(np.float32(R.ref_gdm_distance(x, mu, isr, dim)))
    for order in (1, 2):
        w = rng.standard_normal((5, 17)).astype(np.float32)
        out = np.zeros(17, np.float32)
        R.ref_regression(order, w.reshape(-1), 5, 17, out)
        assert np.array_equal(bits(oracle_regression(w, order=order, right=2, contract=contract)[2]), bits(out))
    amp, w = np.abs(rng.standard_normal(257)).astype(np.float32), rng.uniform(0, 1, 40).astype(np.float32)
    assert bits(np.float32(L.orc_filter_apply(amp, 31, 71, w))) == bits(np.float32(R.ref_filter_apply(amp, 257, 31, 71, w)))
    import math
    for _ in range(300):
        typ, warp, diff = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        d2c = 1.0 / (512 / 16000.0)
        fmaxw = 2595.0 * math.log10(1 + d2c * 256 / 700.0) if warp == 0 else 6.0 * math.asinh(d2c * 256 / 600.0)
        p = [typ, warp, float(rng.uniform(-0.1, 1.1)) * fmaxw, float(rng.uniform(0.01, 0.2)) * fmaxw, 0.0, fmaxw, d2c, diff]
        a, b = _filter_build(R.ref_filter_build, p), _filter_build(L.orc_filter_build, p)
        assert a[:3] == b[:3] and np.array_equal(bits(a[3]), bits(b[3])), p
    for _ in range(300):
        fmax = float(rng.uniform(5, 3000))
        p = [int(rng.integers(0, 3)), float(rng.uniform(0.01, 1.2)) * fmax, float(rng.choice([0.0, 0.05 * fmax])), float(rng.choice([0.5, 2.5 / 3.8])), 0.0, fmax]
        a, b = _boundary(R.ref_filter_boundary, p), _boundary(L.orc_filter_boundary, p)
        assert a[:3] == b[:3] and np.array_equal(bits(a[3]), bits(b[3])), p
    for np1, n_in, n_out, norm in ((0, 40, 40, 0), (1, 22, 20, 1), (0, 9, 3, 1)):
        x = (rng.standard_normal(n_in) * 4).astype(np.float32)
        oa, ob, ta, tb = np.zeros(n_out, np.float32), np.zeros(n_out, np.float32), np.zeros(n_out * n_in, np.float32), np.zeros(n_out * n_in, np.float32)
        R.ref_cosine_transform(np1, n_in, n_out, norm, x, oa, ta)
        L.orc_cosine_transform(np1, n_in, n_out, norm, x, ob, tb)
        assert np.array_equal(bits(oa), bits(ob)) and np.array_equal(bits(ta), bits(tb))
    for f in rng.uniform(0, 8000, 500):
        assert L.orc_bark_derivative(float(f)) == R.ref_bark_derivative(float(f))
    for alpha, n in ((1.0, 9001), (0.97, 9001), (0.9, 3)):
        x = (rng.standard_normal(n) * 1000).astype(np.float32)
        out, y = np.zeros_like(x), x.copy()
        R.ref_preemphasis(alpha, 16000.0, x, n, 4096, -1, out)
        L.orc_preemphasis(y, n, alpha)
        assert np.array_equal(bits(out), bits(y)), (alpha, n)
    for n in (2, 7, 255, 400, 777):
        a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
        assert R.ref_hamming_window(n, a) == 0
        L.orc_hamming_window(b, n)
        assert np.array_equal(bits(a), bits(b)), n
    for dim in (40, 33, 5):
        pdim = (dim + 7) // 8 * 8
        for _ in range(30):
            nk, T = int(rng.integers(1, 12)), int(rng.integers(1, 6))
            ms, xs = np.zeros((nk, pdim), np.float32), np.zeros((T, pdim), np.float32)
            ms[:, :dim], xs[:, :dim] = rng.standard_normal((nk, dim)) * 5, rng.standard_normal((T, dim)) * 5
            cst = (rng.standard_normal(nk) * 10 + 50).astype(np.float32)
            out = np.zeros(T, np.float32)
            R.ref_batch_float_fill(ms.reshape(-1), cst, nk, xs.reshape(-1), T, pdim, out)
            got = np.array([L.orc_batch_float_fill(ms.reshape(-1), cst, nk, np.ascontiguousarray(xs[t]), pdim) for t in range(T)], np.float32)
            assert np.array_equal(bits(got), bits(out))
