"""GPU parity of the GMM scorers in the reference's OTHER arithmetic: amx_gmm_model.tuning contract=fma.

The reference's default build (-march=native, GCC's -ffp-contract=fast) fuses the distance's `sum += df * df` into one fused
multiply-add on any FMA host; about a fifth of the distances then differ in the last bit from the -DMARCH=x86-64 build
(tests/test_contract.py pins both against the reference's own function text compiled both ways).  contract=fma makes every kernel that
evaluates the distance do the same (v_fmac_f32), and the bar stays what it is for contract=off: bit-exact scores and best-density
indices against the oracle library of the SAME build (oracle/liboracle_fma.so), on every path.
"""
import numpy as np
import pytest

from tests import synth
from tests.test_gmm_gpu import _alignment, _cart_adversarial, _screen_worst_case, _tied_adversarial, assert_exact, feats

pytestmark = pytest.mark.gpu
FMA = "contract=fma"


def tun(*items):
    return ",".join((FMA,) + tuple(i for i in items if i))


def test_the_two_contracts_give_different_bits_and_each_matches_its_oracle(ctx):
    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(333, 1, 16, 40, seed=2, pooled=True)
    x = feats(700, 40, 3)
    a, ba = rasr_amd.GmmFeatureScorer(ctx, model).score(x)
    b, bb = rasr_amd.GmmFeatureScorer(ctx, model, tuning=FMA).score(x)
    frac = np.count_nonzero(a.view(np.uint32) != b.view(np.uint32)) / a.size
    assert 0.05 < frac < 0.5, frac                                     # the builds differ ...
    assert np.max(np.abs(a.astype(np.float64) - b) / np.abs(a)) < 1e-6   # ... by rounding only: 1e-4 holds across builds by two orders
    assert np.array_equal(a.argmin(axis=1), b.argmin(axis=1))
    for got, gb, c in ((a, ba, "off"), (b, bb, "fma")):
        osc, obest = OracleGmm(model, contract=c).score(x[:96], mode=0)
        assert np.array_equal(got[:96].view(np.uint32), osc.view(np.uint32)) and np.array_equal(gb[:96], obest)


@pytest.mark.parametrize("n_mix,T", [(1, 1), (7, 31), (17, 256), (45, 257), (333, 700), (1000, 64)])
def test_fused_shapes(ctx, n_mix, T):
    model = synth.gmm_cart(n_mix, 1, 16, 40, seed=400 + n_mix, pooled=True)
    assert_exact(ctx, model, feats(T, 40, 401 + T), tuning=FMA)


@pytest.mark.parametrize("variant", ["", "fused=0", "fused=0,screen_kernel=persist", "fused=0,screen_kernel=simple", "screen=0"])
@pytest.mark.parametrize("dim,pooled", [(16, True), (24, True), (24, False), (32, True), (33, True), (39, True), (40, True), (40, False), (45, True),
                                        (48, False), (64, True)])
def test_every_path_and_dimension(ctx, dim, pooled, variant):
    """fused kernel, the two-kernel screen path with each screen kernel, the evaluate-everything kernel -- every specialised dimension,
    pooled and per-density covariances, twin densities and one-ulp weight neighbours inside the mixtures"""
    model = _cart_adversarial(500 + dim, 70, dim, pooled)
    assert_exact(ctx, model, feats(300, dim, 501), tuning=tun(variant))


@pytest.mark.parametrize("dim", [1, 3, 7, 50, 80])
def test_runtime_dimensions_with_the_scalar_tail(ctx, dim):
    model = synth.gmm_cart(37, 1, 5, dim, seed=20 + dim, pooled=False)
    assert_exact(ctx, model, feats(130, dim, 21), tuning=FMA)


@pytest.mark.parametrize("waves", [8, 12, 16])
@pytest.mark.parametrize("want_best", [True, False])
def test_fused_wave_counts_with_and_without_best_densities(ctx, waves, want_best):
    """every workgroup shape of the fused kernel, in both arithmetics, with AND without the best-density matrix (fused_waves=16 without
    it used to launch the 8-wave kernel on a grid sized for 512 frames per workgroup: half the frames were never scored)"""
    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(70, 1, 16, 40, seed=610, pooled=True)
    x = feats(5000, 40, 611)
    for c, t in (("off", ""), ("fma", FMA)):
        s = rasr_amd.GmmFeatureScorer(ctx, model, tuning=",".join(i for i in (t, "fused_waves=%d" % waves) if i))
        got = s.score(x, want_best=want_best)
        sc = got[0] if want_best else got
        osc, obest = OracleGmm(model, contract=c).score(x, mode=0)
        assert np.array_equal(sc.view(np.uint32), osc.view(np.uint32)), (c, waves, np.abs(sc - osc).max())
        if want_best:
            assert np.array_equal(got[1], obest)


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("dim", [40, 24])
def test_screen_threshold_worst_case(ctx, dim, fused):
    """the constructed worst case of the f16 screen (every rounding error aligned against the true winner): the fused evaluation rounds
    less than the unfused one the threshold was derived for, so the disadvantaged density must still survive and win"""
    import rasr_amd
    from oracle import OracleGmm
    model, x, slots = _screen_worst_case(dim, 96, 900 + dim)
    osc, obest = OracleGmm(model, contract="fma").score(x, mode=0)
    sc, best = rasr_amd.GmmFeatureScorer(ctx, model, tuning=tun("fused=" + fused)).score(x)
    assert np.array_equal(sc.view(np.uint32), osc.view(np.uint32)) and np.array_equal(best, obest)


def test_non_finite_frames_and_operand_range(ctx):
    model = synth.gmm_cart(64, 1, 16, 40, seed=33, pooled=True)
    x = feats(200, 40, 34)
    x[3] = 0.0
    x[5, 2] = np.inf
    x[6, 0] = np.nan
    x[7] = 1e18
    x[8] = 3e3
    x[9] = 7e4          # does not fit the f16 screen operand: every density evaluated
    assert_exact(ctx, model, x, tuning=FMA)


@pytest.mark.parametrize("prune", ["1", "0"])
@pytest.mark.parametrize("kind", ["shared", "partial", "adversarial", "private-cov"])
def test_tied_models(ctx, kind, prune):
    """tied models: gmm_dist_kernel evaluates the distances (the contracted site), the combine / pruned kernels are contraction free"""
    if kind == "shared":
        model, dim = synth.gmm_tied(300, 64, 40, seed=5, pooled=True), 40
    elif kind == "partial":
        model, dim = synth.gmm_tied(200, 128, 24, seed=7, pooled=True, k_per_mix=40), 24
    elif kind == "adversarial":
        model, dim = _tied_adversarial(141, 150, 96, 24, 7), 24
    else:
        model, dim = synth.gmm_tied(120, 64, 40, seed=9, pooled=False), 40
    assert_exact(ctx, model, feats(200, dim, 6), tuning=tun("tied_prune=" + prune))


@pytest.mark.parametrize("dim", [40, 39, 33, 16, 50])
def test_batch_float_scorer(ctx, dim):
    """batch-diagonal-maximum-float: _mm_add_ps(s, _mm_mul_ps(x, x)) contracts like the scalar form"""
    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(150, 1, 9, dim, seed=60 + dim, pooled=True)
    x = feats(300, dim, 61)
    got = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="batch-diagonal-maximum-float", tuning=FMA).score(x, want_best=False)
    want = OracleGmm(model, contract="fma").score_batch_float(x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    other = OracleGmm(model, contract="off").score_batch_float(x)
    assert np.count_nonzero(got.view(np.uint32) != other.view(np.uint32)) > 0


@pytest.mark.parametrize("tied", [False, True])
def test_log_add_scorer(ctx, tied):
    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_tied(100, 32, 40, seed=15) if tied else synth.gmm_cart(100, 1, 8, 40, seed=14, pooled=False)
    x = feats(150, 40, 16)
    sc, best = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="diagonal-sum", tuning=FMA).score(x)
    osc, obest = OracleGmm(model, contract="fma").score(x, mode=1)
    assert np.allclose(sc, osc, rtol=1e-5, atol=1e-5), np.abs(sc - osc).max()
    assert np.array_equal(best, obest)


@pytest.mark.parametrize("kind", ["cart", "cart-wide", "tied"])
def test_statistics_entry_points(ctx, kind):
    """amx_gmm_score_stats_dev / _u8_dev / amx_gmm_best_density_dev in fma mode: the same bits as amx_gmm_score_dev in fma mode, which
    the tests above hold to the oracle"""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    if kind == "cart":
        model = synth.gmm_cart(333, 1, 16, 40, seed=101, pooled=True)
    elif kind == "cart-wide":
        model = synth.gmm_cart(40, 10, 24, 40, seed=102, pooled=False)
    else:
        model = synth.gmm_tied(120, 64, 40, seed=103)
    T, M = 1000, len(model["mix_offsets"]) - 1
    x = feats(T, 40, 104)
    sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning=FMA)
    ref_scores, ref_best = sc.score(x)
    osc, obest = OracleGmm(model, contract="fma").score(x[:40], mode=0)
    assert np.array_equal(ref_scores[:40].view(np.uint32), osc.view(np.uint32)) and np.array_equal(ref_best[:40], obest)
    xd = torch.from_numpy(x).cuda()
    scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
    state = torch.empty((T,), dtype=torch.int32, device="cuda")
    counts = torch.zeros((M,), dtype=torch.int64, device="cuda")
    ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
    ctx.use_torch_stream()
    kmax = int(np.diff(model["mix_offsets"]).max())
    for dt in (torch.int32,) + ((torch.uint8,) if kmax <= 254 else ()):
        bestd = torch.empty((T, M), dtype=dt, device="cuda")
        counts.zero_()
        sc.score_stats_dev(xd, T, scores, bestd, state, counts, ssum)
        torch.cuda.synchronize()
        assert np.array_equal(scores.cpu().numpy().view(np.uint32), ref_scores.view(np.uint32))
        assert np.array_equal(bestd.cpu().numpy().astype(np.uint32), ref_best)
        assert np.array_equal(state.cpu().numpy(), ref_scores.argmin(axis=1))
    mix = np.random.Generator(np.random.PCG64(725)).integers(0, M, T).astype(np.int32)
    md = torch.from_numpy(mix).cuda()
    bd = torch.full((T,), 7, dtype=torch.int32, device="cuda")
    sd = torch.zeros((T,), dtype=torch.float32, device="cuda")
    sc.best_density_dev(xd, T, md, bd, sd)
    torch.cuda.synchronize()
    assert np.array_equal(bd.cpu().numpy().astype(np.uint32), ref_best[np.arange(T), mix])
    assert np.array_equal(sd.cpu().numpy().view(np.uint32), ref_scores[np.arange(T), mix].view(np.uint32))


def test_baum_welch_accumulators(ctx):
    import torch

    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(30, 1, 9, 40, seed=80, pooled=True)
    T, n_mix = 1500, 30
    x = (feats(T, 40, 84) * 0.6).astype(np.float32)
    w = np.random.Generator(np.random.PCG64(85)).uniform(0.2, 1.0, T)
    mix = _alignment(T, n_mix, 86)
    sc, o = rasr_amd.GmmFeatureScorer(ctx, model, tuning=FMA), OracleGmm(model, contract="fma")
    ctx.use_torch_stream()
    xd, md, wd = (torch.from_numpy(a).cuda() for a in (x, mix.astype(np.int32), w))
    acc = torch.zeros(sc.accumulator_size(), dtype=torch.float64, device="cuda")
    sc.accumulate_weighted_dev(rasr_amd.AMX_GMM_BAUM_WELCH, xd, T, md, wd, None, 0, acc)
    torch.cuda.synchronize()
    got, want = acc.cpu().numpy(), o.accumulate_weighted(1, x, mix, w)
    assert np.allclose(got, want, rtol=2e-5, atol=2e-6 * max(np.abs(want).max(), 1.0)), np.abs(got - want).max()


def test_full_size_shard_is_bit_exact_on_a_sample(ctx):
    """BASELINE config 5's GMM leg (10 000 states x 16 densities, 63 936 frames) in fma mode: every row of a frame sample against the
    oracle of the same build, arg-min state on all of them"""
    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(10000, 16, 16, 40, seed=31, pooled=True)
    x = feats(63936, 40, 32)
    sc, best = rasr_amd.GmmFeatureScorer(ctx, model, tuning=FMA).score(x)
    rows = np.array([0, 1, 31, 32, 255, 256, 383, 384, 20000, 63935])
    osc, obest = OracleGmm(model, contract="fma").score(x[rows], mode=0)
    assert np.array_equal(sc[rows].view(np.uint32), osc.view(np.uint32)) and np.array_equal(best[rows], obest)


def test_what_contract_fma_refuses(ctx):
    """only the specialised-wave lab kernel refuses (round 6: every scorer type takes contract=fma -- tests/test_contract_gpu.py holds them
    to the oracle of that build); a value that is not a contract is an error, not the default"""
    import rasr_amd
    model = synth.gmm_cart(20, 1, 8, 40, seed=1, pooled=True)
    x = feats(10, 40, 2)
    with pytest.raises(rasr_amd.AmxError) as e:
        rasr_amd.GmmFeatureScorer(ctx, model, tuning="contract=fma,fused_waves=13")
    assert e.value.status == -2
    for typ in ("SIMD-diagonal-maximum", "batch-diagonal-maximum-int", "preselection-batch-float", "preselection-batch-int"):
        s = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type=typ, tuning=FMA)
        if typ.startswith("preselection"):
            s.set_preselection(8, 4, 3, 40000.0)
        assert np.isfinite(s.score(x, want_best=False)).all(), typ
    for bad in ("contract=fmaa", "contract=1", "contract=", "fused_waves=10", "chunk=abc", "chunk=-5", "screen_kernel=rowz", "fr=3"):
        with pytest.raises(rasr_amd.AmxError) as e:
            rasr_amd.GmmFeatureScorer(ctx, model, tuning=bad)
        assert e.value.status == -1, bad
