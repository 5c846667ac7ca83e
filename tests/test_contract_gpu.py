"""contract=off | fma OUTSIDE the GMM float scorers (round 6; tests/test_gmm_contract_gpu.py holds those): the context-wide setting
(amx_set_contract) and the entry points it reaches -- signal-regression, signal-matrix-multiplication-f32, the amplitude-spectrum-energy
normalisation, the gammatone front end (design, cascade, both integrations, cosine transform), the preselection scorer (clustering
distances and scores), the quantised scorers (integer arithmetic: they accept the mode), the MFCC tables -- bit for bit against the
oracle library of the SAME build of the reference (oracle/liboracle.so | liboracle_fma.so), and different from the other build's
where the reference's own two builds differ (profiles/r05/contract_pins.json)."""
import numpy as np
import pytest

from tests import synth
from tests.test_backend import _plan, seg

pytestmark = pytest.mark.gpu

CONTRACTS = ("off", "fma")


@pytest.fixture()
def cctx(ctx):
    """the session context, handed out in contract=off and restored to it (other tests expect the default)"""
    ctx.set_contract("off")
    yield ctx
    ctx.set_contract("off")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_context_contract_round_trip(cctx):
    import rasr_amd
    assert cctx.contract() == "off"                       # the library's default: RASR with -DMARCH=x86-64
    cctx.set_contract("fma")
    assert cctx.contract() == "fma"
    assert b"default build" in cctx.L.amx_contract_description(1) and b"x86-64" in cctx.L.amx_contract_description(0)
    with pytest.raises(rasr_amd.AmxError):
        cctx.set_contract(2)
    assert cctx.contract() == "fma"                       # a refused value changes nothing


@pytest.mark.parametrize("contract", CONTRACTS)
def test_regression_and_matrix_multiply_follow_the_contract(cctx, contract):
    """Signal/Regression.cc:24-65 and Math::Vector's dot product (Math/Vector.hh:94-101) -- 29 % and 65 % of the outputs differ between
    the reference's two builds; the kernels fuse exactly the oracle's ORC_FMAF sites"""
    import torch
    from oracle.binding import oracle_matrix_multiply, oracle_regression
    cctx.set_contract(contract)
    other = "fma" if contract == "off" else "off"
    lens = [1, 2, 5, 9, 64, 333, 7]
    plan = _plan(cctx, lens)
    F, dim, ld = sum(lens), 13, 20
    x = np.zeros((F, ld), np.float32)
    x[:, 3:3 + dim] = seg(F, dim, 11)
    xd = torch.from_numpy(x).cuda()
    cctx.use_torch_stream()
    off = np.concatenate([[0], np.cumsum(lens)])
    differs = 0
    for order in (1, 2):
        for right in (1, 2, 4):
            out = torch.zeros((F, dim), dtype=torch.float32, device="cuda")
            cctx.regression(plan, xd[:, 3:], ld, dim, out, dim, order=order, right=right)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            for i in range(len(lens)):
                s = x[off[i]:off[i + 1], 3:3 + dim]
                assert np.array_equal(bits(got[off[i]:off[i + 1]]), bits(oracle_regression(s, order, right, contract=contract))), (order, right, i)
                differs += int((bits(got[off[i]:off[i + 1]]) != bits(oracle_regression(s, order, right, contract=other))).sum())
    assert differs > 0                                    # the two builds are two arithmetics
    M = np.random.Generator(np.random.PCG64(12)).standard_normal((45, dim)).astype(np.float32)
    out = torch.zeros((F, 48), dtype=torch.float32, device="cuda")
    cctx.matrix_multiply(torch.from_numpy(M).cuda(), 45, dim, xd[:, 3:], ld, F, out, 48)
    torch.cuda.synchronize()
    got = out.cpu().numpy()[:, :45]
    assert np.array_equal(bits(got), bits(oracle_matrix_multiply(M, x[:, 3:3 + dim], contract=contract)))
    assert (bits(got) != bits(oracle_matrix_multiply(M, x[:, 3:3 + dim], contract=other))).any()


@pytest.mark.parametrize("contract", CONTRACTS)
def test_vector_normalisations_follow_the_contract(cctx, contract):
    """amplitude-spectrum-energy has the one contracted site (front * front + back * back); the other five are the same in both builds"""
    import torch
    from oracle.binding import oracle_vector_normalize
    cctx.set_contract(contract)
    cctx.use_torch_stream()
    for kind in ("amplitude-spectrum-energy", "energy", "maximum", "mean-energy", "mean", "variance"):
        for n, dim in ((300, 40), (77, 257), (1000, 13), (5, 2)):
            x = seg(n, dim, 40 + dim)
            xd = torch.from_numpy(x).cuda()
            out = torch.empty((n, dim), dtype=torch.float32, device="cuda")
            cctx.vector_normalize(kind, xd, dim, n, dim, out, dim)
            torch.cuda.synchronize()
            assert np.array_equal(bits(out.cpu().numpy()), bits(oracle_vector_normalize(x, kind, contract=contract))), (kind, n, dim)


@pytest.mark.parametrize("contract", CONTRACTS)
@pytest.mark.parametrize("kw", [dict(), dict(cf_mode=1, channels=70, cascade=2), dict(channels=68, max_freq=7500.0, si_length=9, si_shift=4, power=0.1, n_ceps=12),
                                dict(sample_rate=8000.0, max_freq=3800.0, warp_freq_break=3300.0, ti_length_s=0.032, ti_shift_s=0.004)])
def test_gammatone_follows_the_contract(cctx, contract, kw):
    """design (warping, centre frequencies), every sample of the filter cascade (27 342 of 30 100 outputs differ between the reference's
    builds), temporal integration (f64 fused), spectral integration and cosine transform: bit-identical to the oracle of the build; the
    mode comes from the context, or from the handle's own tuning string against the context's"""
    import torch

    import rasr_amd
    from oracle.binding import GammatoneCfg, OracleGammatone
    cctx.set_contract(contract)
    fe = rasr_amd.GammatoneExtractor(cctx, **kw)
    o = OracleGammatone(GammatoneCfg.default(**kw), contract=contract)
    cf, co = fe.tables()
    assert np.array_equal(bits(cf), bits(o.center_frequencies)) and np.array_equal(bits(co), bits(o.coefficients))
    cctx.set_contract("off")
    override = rasr_amd.GammatoneExtractor(cctx, tuning="contract=" + contract, **kw)   # the handle's string wins over the context
    cctx.use_torch_stream()
    any_diff = False
    for n in (2, 401, 5281, 16000):
        pcm = synth.waveform(n, seed=40 + n)
        want, wfilt = o.run(pcm, want_filtered=True)
        got = fe.run(pcm)
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (n, np.abs(got - want).max())
        assert np.array_equal(bits(override.run(pcm)), bits(want))
        pd = torch.from_numpy(pcm).cuda()
        out = torch.empty(want.shape, dtype=torch.float32, device="cuda")
        filt = torch.empty(wfilt.shape, dtype=torch.float32, device="cuda")
        fe.run_batch_dev([0, n], pd, out, filt)
        torch.cuda.synchronize()
        assert np.array_equal(bits(filt.cpu().numpy()), bits(wfilt)) and np.array_equal(bits(out.cpu().numpy()), bits(want))
        if n == 16000:
            oth = OracleGammatone(GammatoneCfg.default(**kw), contract="fma" if contract == "off" else "off").run(pcm)
            any_diff = bool((bits(oth) != bits(want)).any())
    assert any_diff
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.GammatoneExtractor(cctx, tuning="contract=maybe")


@pytest.mark.parametrize("contract", CONTRACTS)
@pytest.mark.parametrize("n_mix,kmax,dim,clusters,select", [(200, 16, 40, 64, 8), (50, 8, 24, 16, 4), (300, 12, 33, 256, 32), (120, 16, 39, 32, 32), (60, 8, 20, 16, 5)])
def test_preselection_float_follows_the_contract(cctx, contract, n_mix, kmax, dim, clusters, select):
    """Mm::unrolledVectorDistance (score += df * df: one vfmadd231ss per term in the default build, read off libref_native.so) under the
    clustering and the cluster selection, and the batch-float distance under the scores: clustering, cluster means and scores equal the
    oracle's of the build; the mode reaches the scorer through the CONTEXT here (no tuning string)"""
    import rasr_amd
    from oracle import OracleGmm
    from tests.test_gmm_gpu import feats
    cctx.set_contract(contract)
    model = synth.gmm_cart(n_mix, 1, kmax, dim, seed=500 + n_mix, pooled=True)
    x = feats(333, dim, 501)
    sc = rasr_amd.GmmFeatureScorer(cctx, model, feature_scorer_type="preselection-batch-float")
    sc.set_preselection(clusters, select, 5, 40000.0)
    got = sc.score(x, want_best=False)
    want, wcof, wcm = OracleGmm(model, contract=contract).score_preselection_float(x, clusters, select, 5, 40000.0)
    cof, cm = sc.preselection_clustering()
    assert np.array_equal(cof, wcof) and np.array_equal(bits(cm), bits(wcm[:, :dim]))
    assert np.array_equal(bits(got), bits(want)), np.abs(got - want).max()
    plain = rasr_amd.GmmFeatureScorer(cctx, model).score(x, want_best=False)          # diagonal-maximum created under the same context
    assert np.array_equal(bits(plain), bits(OracleGmm(model, contract=contract).score(x, mode=0)[0]))


@pytest.mark.parametrize("contract", CONTRACTS)
def test_quantised_scorers_accept_the_contract(cctx, contract):
    """SIMD-diagonal-maximum, batch-diagonal-maximum-int and preselection-batch-int: integer distances; their one f64 site
    (gaussLogNormFactor) follows the contract on the host.  Bit-exact against the oracle of the build, no AMX_ERR_UNSUPPORTED"""
    import rasr_amd
    from oracle import OracleGmm
    from tests.test_gmm_gpu import feats
    cctx.set_contract(contract)
    for dim, n_mix in ((40, 120), (33, 70), (20, 40)):
        model = synth.gmm_cart(n_mix, 1, 12, dim, seed=700 + dim, pooled=True)
        x = feats(200, dim, 701)
        o = OracleGmm(model, contract=contract)
        s, b = rasr_amd.GmmFeatureScorer(cctx, model, feature_scorer_type="SIMD-diagonal-maximum").score(x)
        ws, wb, _ = o.score_simd(x)
        assert np.array_equal(bits(s), bits(ws)) and np.array_equal(b, wb)
        s = rasr_amd.GmmFeatureScorer(cctx, model, feature_scorer_type="batch-diagonal-maximum-int").score(x, want_best=False)
        assert np.array_equal(bits(s), bits(o.score_batch_int(x)))
        p = rasr_amd.GmmFeatureScorer(cctx, model, feature_scorer_type="preselection-batch-int")
        p.set_preselection(16, 4, 5, 0.0)
        ws, wcof, wcm = o.score_preselection_int(x, 16, 4, 5)
        assert np.array_equal(bits(p.score(x, want_best=False)), bits(ws))


@pytest.mark.parametrize("contract", CONTRACTS)
def test_mfcc_tables_follow_the_contract(cctx, contract):
    """the filter bank's geometry (f64: coverage, stretch-to-cover centres, setStart / setEnd, the include-boundary count, the bark
    derivative) as the build of the reference contracts it: window, filter bounds, weights and cosine table bit-identical to the oracle's;
    the cepstra stay inside the front end's bar against it (MFCC 1e-4 |ref| + 1e-4; the PLP family's band of tests/test_mfcc_gpu.py)"""
    import rasr_amd
    from oracle.binding import MfccCfg, OracleMfcc
    from tests.test_mfcc_gpu import PLP_ATOL, PLP_RTOL
    cctx.set_contract(contract)
    pcm = synth.waveform(16000, seed=3)
    cases = [(lambda: rasr_amd.MfccExtractor(cctx, nr_cepstrum_coefficients=40, filter_width=138.0), MfccCfg.default(n_ceps=40, filter_width=138.0), 1e-4, 1e-4),
             (lambda: rasr_amd.MfccExtractor(cctx, nr_cepstrum_coefficients=16), MfccCfg.default(n_ceps=16), 1e-4, 1e-4),
             (lambda: rasr_amd.MfccExtractor(cctx, nr_cepstrum_coefficients=12, alpha=0.97), MfccCfg.default(n_ceps=12, alpha=0.97), 1e-4, 1e-4),
             (lambda: rasr_amd.MfccExtractor.plp(cctx), MfccCfg.plp(), PLP_RTOL, PLP_ATOL),
             (lambda: rasr_amd.MfccExtractor.plp(cctx, sample_rate=8000.0, spacing=0.973442), MfccCfg.plp(spacing=0.973442, sample_rate=8000.0), PLP_RTOL, PLP_ATOL),
             (lambda: rasr_amd.MfccExtractor(cctx, nr_cepstrum_coefficients=13, front_end="mfplp", nr_autocorrelation_coefficients=13, normalize=True),
              MfccCfg.mfplp(n_ceps=13, n_autocorrelation=13), PLP_RTOL, PLP_ATOL)]
    for make, ocfg, rtol, atol in cases:
        fe, o = make(), OracleMfcc(ocfg, contract=contract)
        t = fe.tables()
        s_, e_, o_, w_ = o.filters
        assert np.array_equal(bits(t["window"]), bits(o.window))
        assert np.array_equal(t["filter_start"], s_) and np.array_equal(t["filter_end"], e_) and np.array_equal(t["filter_offset"], o_)
        assert np.array_equal(bits(t["filter_weights"]), bits(w_)) and np.array_equal(bits(t["dct"]), bits(o.dct))
        got, want = fe.run(pcm), o.run(pcm)
        ok = np.isfinite(want)
        assert np.all(np.abs(got - want)[ok] <= rtol * np.abs(want)[ok] + atol), np.abs(got - want)[ok].max()
    # the handle's own string beats the context, an unknown value fails the creation
    cctx.set_contract("off")
    fe = rasr_amd.MfccExtractor.plp(cctx, tuning="contract=" + contract)
    assert np.array_equal(bits(fe.tables()["filter_weights"]), bits(OracleMfcc(MfccCfg.plp(), contract=contract).filters[3]))
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.MfccExtractor(cctx, tuning="contract=native")
