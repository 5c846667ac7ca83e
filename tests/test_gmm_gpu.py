"""GPU parity: diagonal-GMM scorer through the C ABI against the oracle (bit-exact in max mode)."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def feats(T, dim, seed):
    return np.random.Generator(np.random.PCG64(seed)).standard_normal((T, dim)).astype(np.float32)


def contract_of(tuning):
    """which build of the reference a tuning string asks for: "contract=fma" -> "fma", else "off" (the oracle library to compare with)"""
    return "fma" if "contract=fma" in (tuning or "") else "off"


def assert_exact(ctx, model, x, **kw):
    import rasr_amd
    from oracle import OracleGmm
    sc, best = rasr_amd.GmmFeatureScorer(ctx, model, **kw).score(x)
    osc, obest = OracleGmm(model, contract=contract_of(kw.get("tuning")),
                           **{k: v for k, v in kw.items() if k not in ("feature_scorer_type", "tuning")}).score(x, mode=0)
    assert np.array_equal(sc.view(np.uint32), osc.view(np.uint32)), np.abs(sc - osc).max()
    assert np.array_equal(best, obest)


def test_known_answer(ctx):
    """d=4, pooled var 2, means 0/1, weights .25/.75, x=0.5 -> 5.59973 (reference output)."""
    import rasr_amd
    model = dict(dim=4, mix_offsets=np.array([0, 2], np.uint32), dens_index=np.array([0, 1], np.uint32),
                 log_weight=np.log(np.array([0.25, 0.75])), dens_mean=np.array([0, 1], np.uint32),
                 dens_cov=np.array([0, 0], np.uint32), means=np.array([[0] * 4, [1] * 4], np.float32),
                 variances=np.full((1, 4), 2, np.float32))
    sc, best = rasr_amd.GmmFeatureScorer(ctx, model).score(np.full((1, 4), 0.5, np.float32))
    assert abs(sc[0, 0] - 5.59973) < 1e-5 and best[0, 0] == 1


@pytest.mark.parametrize("pooled", [True, False])
@pytest.mark.parametrize("T", [1, 63, 64, 65, 256, 1000])
def test_cart_model_exact(ctx, pooled, T):
    model = synth.gmm_cart(256, 1, 8, 40, seed=2, pooled=pooled)   # BASELINE config 1 model
    assert_exact(ctx, model, feats(T, 40, 4))


@pytest.mark.parametrize("dim", [1, 3, 4, 7, 16, 33, 39, 45, 50, 64, 80])
def test_dimensions_including_sse_tail(ctx, dim):
    """dim % 4 != 0 exercises the scalar tail of the reference's SSE distance; dims without a
    specialised kernel take the runtime-dimension path."""
    model = synth.gmm_cart(37, 1, 5, dim, seed=20 + dim, pooled=False)
    assert_exact(ctx, model, feats(130, dim, 21))


def test_scales(ctx):
    model = synth.gmm_cart(50, 2, 6, 40, seed=9, pooled=False)
    assert_exact(ctx, model, feats(100, 40, 10), mixture_weight_scale=0.7, gaussian_scale=1.3)


def test_ties_first_minimum_wins(ctx):
    """duplicate densities: the reference's strict '>' keeps the first of equal scores"""
    model = synth.gmm_cart(20, 4, 4, 40, seed=12, pooled=True)
    model["means"][1::4] = model["means"][0::4]          # density 1 of each mixture == density 0
    lw = model["log_weight"].reshape(20, 4)
    lw[:, 1] = lw[:, 0]
    model["log_weight"] = lw.reshape(-1)
    assert_exact(ctx, model, feats(64, 40, 13))


@pytest.mark.parametrize("pooled", [True, False])
def test_tied_mixture_two_stage_exact(ctx, pooled):
    model = synth.gmm_tied(300, 64, 40, seed=5, pooled=pooled)      # 300*64 entries >> 64 densities
    assert_exact(ctx, model, feats(200, 40, 6))


def test_tied_mixture_partial_lists(ctx):
    model = synth.gmm_tied(200, 128, 24, seed=7, pooled=True, k_per_mix=40)
    assert_exact(ctx, model, feats(77, 24, 8))


@pytest.mark.parametrize("tied", [False, True])
def test_log_add_scorer(ctx, tied):
    """diagonal-sum: f32 throughout; the device uses a single-pass online log-sum-exp, the reference a
    two-pass one, and expf/logf differ by ulps -> 1e-5 relative (requirement: 1e-4)."""
    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_tied(100, 32, 40, seed=15) if tied else synth.gmm_cart(100, 1, 8, 40, seed=14, pooled=False)
    x = feats(150, 40, 16)
    sc, best = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="diagonal-sum").score(x)
    osc, obest = OracleGmm(model).score(x, mode=1)
    assert np.allclose(sc, osc, rtol=1e-5, atol=1e-5), np.abs(sc - osc).max()
    assert np.array_equal(best, obest)


def test_argmin_state_exact_at_scale(ctx):
    """10 000 states (BASELINE config 3, CART-style instance, reduced K for oracle time): argmin state per
    frame is identical; scores bit-exact on a frame sample."""
    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(10000, 2, 2, 40, seed=31, pooled=True)
    x = feats(256, 40, 32)
    sc, _ = rasr_amd.GmmFeatureScorer(ctx, model).score(x)
    osc, _ = OracleGmm(model).score(x[:16], mode=0)
    assert np.array_equal(sc[:16].view(np.uint32), osc.view(np.uint32))
    assert np.array_equal(sc[:16].argmin(axis=1), osc.argmin(axis=1))


def test_errors(ctx):
    import rasr_amd
    model = synth.gmm_cart(4, 1, 2, 8, seed=1)
    bad = dict(model)
    bad["variances"] = model["variances"].copy()
    bad["variances"][0, 0] = 0.0
    with pytest.raises(rasr_amd.AmxError) as e:
        rasr_amd.GmmFeatureScorer(ctx, bad)
    assert e.value.status == -1
    s = rasr_amd.GmmFeatureScorer(ctx, model)
    sc, _ = s.score(np.zeros((0, 8), np.float32))
    assert sc.shape == (0, 4)


@pytest.mark.parametrize("dim", [40, 39, 33, 16])
def test_batch_float_scorer_exact(ctx, dim):
    """batch-diagonal-maximum-float: bit-identical to the oracle's restatement of Mm::BatchFloatFeatureScorer
    (8-wide blocks incl. dims that are not a multiple of 8), and within f32 round-off of diagonal-maximum"""
    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(300, 1, 8, dim, seed=40 + dim, pooled=True)
    x = feats(200, dim, 41)
    got = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="batch-diagonal-maximum-float").score(x, want_best=False)
    o = OracleGmm(model)
    want = o.score_batch_float(x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    assert np.allclose(got, o.score(x)[0], rtol=3e-6)


@pytest.mark.parametrize("contract", ["off", "fma"])
def test_batch_float_scorer_minimum_is_the_references_min_ps(ctx, contract):
    """Mm::BatchFloatFeatureScorer takes the minimum as _mm_min_ps(score, s) = (score < s ? score : s): a NaN sum REPLACES the score and a
    later sum replaces the NaN (Mm/BatchFeatureScorer.cc:245, pinned on the reference's function text in tests/test_contract.py).
    Frames with NaN -> NaN scores; a mixture whose FIRST density has an infinite mean, on a frame that is infinite there (inf - inf): NaN,
    then +inf from the other densities -> +inf (min(s, score) would have kept FLT_MAX); the LAST density infinite: NaN"""
    import rasr_amd
    from oracle import OracleGmm
    dim = 40
    model = synth.gmm_cart(60, 1, 6, dim, seed=77, pooled=True)
    off, idx = model["mix_offsets"], model["dens_index"]
    means = model["means"].copy()
    first = int(model["dens_mean"][idx[off[5]]])          # first density of mixture 5
    last = int(model["dens_mean"][idx[off[9 + 1] - 1]])   # last density of mixture 9
    means[first, 3] = np.inf
    means[last, 7] = np.inf
    model["means"] = means
    x = feats(64, dim, 78)
    x[10, 0] = np.nan
    x[20, 3] = np.inf
    x[30, 7] = np.inf
    tun = "contract=" + contract
    got = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="batch-diagonal-maximum-float", tuning=tun).score(x, want_best=False)
    want = OracleGmm(model, contract=contract).score_batch_float(x)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), np.argwhere(~same)[:5]
    assert np.isnan(got[10]).all() and np.isposinf(got[20, 5]) and np.isnan(got[30, 9])
    # the preselection scorer runs the same loop (Mm/BatchFeatureScorer.cc:306-315): with every cluster selected and a finite model, a NaN
    # frame scores NaN -- `if (s == max) s = backoff` does not catch it -- and a frame that is infinite in one component scores the back-off
    if contract == "off":
        clean = synth.gmm_cart(60, 1, 6, dim, seed=77, pooled=True)
        sc = rasr_amd.GmmFeatureScorer(ctx, clean, feature_scorer_type="preselection-batch-float")
        sc.set_preselection(16, 16, 3, 777.0)
        g2 = sc.score(x, want_best=False)
        w2, _, _ = OracleGmm(clean).score_preselection_float(x, 16, 16, 3, 777.0)
        same = (g2.view(np.uint32) == w2.view(np.uint32)) | (np.isnan(g2) & np.isnan(w2))
        assert same.all(), np.argwhere(~same)[:5]
        assert np.isnan(g2[10]).all() and (g2[20] == 777.0).all()


def test_batch_float_scorer_errors(ctx):
    import rasr_amd
    s = rasr_amd.GmmFeatureScorer(ctx, synth.gmm_cart(10, 1, 3, 40, seed=1, pooled=False), feature_scorer_type="batch-diagonal-maximum-float")
    with pytest.raises(rasr_amd.AmxError) as e:
        s.score(feats(4, 40, 2), want_best=False)
    assert e.value.status == -1 and "globally pooled covariance" in str(e.value)


@pytest.mark.parametrize("pooled", [True, False])
def test_viterbi_accumulators(ctx, pooled):
    """GMM training statistics: f64 sums equal the oracle's sequential accumulation up to summation order (1e-12),
    weights exactly; accumulating twice doubles everything (buffers add, which is what the epoch all-reduce relies on)."""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(50, 1, 6, 40, seed=60, pooled=pooled)
    x = feats(3000, 40, 61)
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    o = OracleGmm(model)
    xd = torch.from_numpy(x).cuda()
    scores = torch.empty((3000, 50), dtype=torch.float32, device="cuda")
    best = torch.empty((3000, 50), dtype=torch.int32, device="cuda")
    ctx.use_torch_stream()
    sc.score_dev(xd, 3000, scores, best)
    mix = scores.argmin(dim=1).to(torch.int32)           # Viterbi "alignment" = best state per frame
    acc = torch.zeros(sc.accumulator_size(), dtype=torch.float64, device="cuda")
    sc.accumulate_dev(xd, 3000, mix, best, 50, acc)
    torch.cuda.synchronize()
    got = acc.cpu().numpy()
    osc, obest = o.score(x)
    omix = osc.argmin(axis=1).astype(np.uint32)
    assert np.array_equal(mix.cpu().numpy().astype(np.uint32), omix)
    want = o.accumulate(x, omix, obest[np.arange(3000), omix])
    assert sc.accumulator_size() == o.accumulator_size() == len(want)
    nk, nm, nc = int(model["mix_offsets"][-1]), model["means"].shape[0], model["variances"].shape[0]
    assert np.array_equal(got[:nk], want[:nk]) and got[:nk].sum() == 3000
    assert np.array_equal(got[nk:nk + nm], want[nk:nk + nm])
    assert np.allclose(got, want, rtol=1e-12, atol=1e-9)
    sc.accumulate_dev(xd, 3000, mix, best, 50, acc)
    torch.cuda.synchronize()
    assert np.allclose(acc.cpu().numpy(), 2 * want, rtol=1e-12, atol=1e-9)
    # per-frame density list (best_density_ld == 0)
    acc2 = torch.zeros_like(acc)
    chosen = best[torch.arange(3000, device="cuda"), mix.long()].contiguous()
    sc.accumulate_dev(xd, 3000, mix, chosen, 0, acc2)
    torch.cuda.synchronize()
    assert np.allclose(acc2.cpu().numpy(), want, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("pooled", [True, False])
def test_accumulators_skip_frames_without_a_density(ctx, pooled):
    """a NaN frame has no best density (0xffffffff / 0xff: what the scorers and amx_gmm_best_density_dev write), a mixture index may lie
    outside the model, an index may point behind its mixture's last density: such frames contribute NOTHING to the statistics -- in the
    u32 matrix form, the byte matrix form, the per-frame list (best_density_ld = 0) and the weighted kernel.  (The sentinel used to be
    added to mix_off[m]: the frame went to the previous mixture's last density, or 4 G entries out of bounds for mixture 0.)"""
    import torch

    import rasr_amd
    model = synth.gmm_cart(40, 2, 6, 24, seed=160, pooled=pooled)
    T, M = 700, 40
    x = feats(T, 24, 161)
    x[5, 3] = np.nan          # frame 5: no density beats FLT_MAX
    x[300] = np.inf
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    xd = torch.from_numpy(x).cuda()
    scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
    best = torch.empty((T, M), dtype=torch.int32, device="cuda")
    ctx.use_torch_stream()
    sc.score_dev(xd, T, scores, best)
    mix = np.random.Generator(np.random.PCG64(162)).integers(0, M, T).astype(np.int32)
    mix[0] = 0                # mixture 0 with the sentinel: the old code read k_dens[0xffffffff]
    mix[17] = M               # outside the model
    mix[18] = -1
    md = torch.from_numpy(mix).cuda()
    bh = best.cpu().numpy().astype(np.uint32)
    assert (bh[5] == 0xffffffff).all() and (bh[300] == 0xffffffff).all()
    bh[0, 0] = 0xffffffff
    bh[40, mix[40]] = 6       # behind the last density of a mixture with fewer than seven
    n_of = np.diff(model["mix_offsets"])
    if n_of[mix[40]] > 6:
        bh[40, mix[40]] = 0xfffffffe
    skip = np.zeros(T, bool)
    skip[[0, 5, 17, 18, 40, 300]] = True
    # what the statistics must be: the oracle over the frames that do have a density
    from oracle import OracleGmm
    keep = np.nonzero(~skip)[0]
    want = OracleGmm(model).accumulate(x[keep], mix[keep].astype(np.uint32), bh[keep, mix[keep]])
    nk = int(model["mix_offsets"][-1])
    full = torch.from_numpy(bh.astype(np.int32)).cuda()
    acc = torch.zeros(sc.accumulator_size(), dtype=torch.float64, device="cuda")
    sc.accumulate_dev(xd, T, md, full, M, acc)                                             # u32 matrix
    torch.cuda.synchronize()
    got = acc.cpu().numpy()
    assert np.isfinite(got).all() and got[:nk].sum() == len(keep) and np.array_equal(got[:nk], want[:nk])
    assert np.allclose(got, want, rtol=1e-12, atol=1e-9)
    b8 = torch.from_numpy(np.minimum(bh, 255).astype(np.uint8)).cuda()                    # byte matrix (0xff = no density)
    acc.zero_()
    sc.accumulate_dev(xd, T, md, b8, M, acc)
    torch.cuda.synchronize()
    assert np.allclose(acc.cpu().numpy(), want, rtol=1e-12, atol=1e-9) and acc.cpu().numpy()[:nk].sum() == len(keep)
    safe = np.clip(mix, 0, M - 1)
    per = torch.from_numpy(np.where((mix >= 0) & (mix < M), bh[np.arange(T), safe], 0).astype(np.int32)).cuda()   # per-frame list
    acc.zero_()
    sc.accumulate_dev(xd, T, md, per, 0, acc)
    torch.cuda.synchronize()
    assert np.allclose(acc.cpu().numpy(), want, rtol=1e-12, atol=1e-9)
    w = torch.ones(T, dtype=torch.float64, device="cuda")                                  # weighted Viterbi, unit weights
    acc.zero_()
    sc.accumulate_weighted_dev(rasr_amd.AMX_GMM_VITERBI, xd, T, md, w, full, M, acc)
    torch.cuda.synchronize()
    assert np.allclose(acc.cpu().numpy(), want, rtol=1e-12, atol=1e-9)


def _tied_adversarial(seed, n_mix, n_dens, dim, dup_every, big=False):
    """uniform-list tied model whose densities contain exact duplicates (equal f64 sums: the FIRST must win) and whose
    weights contain neighbours one f32 ulp apart (sums that round to the same f32: the reference's index then moves to
    the LAST density that is still smaller in f64)."""
    model = synth.gmm_tied(n_mix, n_dens, dim, seed=seed, pooled=True, alpha=2.0)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    means = model["means"]
    for d in range(dup_every, n_dens, dup_every):
        means[d] = means[d - dup_every]                 # same distance for every frame
    lw = model["log_weight"].reshape(n_mix, n_dens)
    for d in range(dup_every, n_dens, dup_every):
        which = rng.integers(0, 3, n_mix)
        base = (-2 * lw[:, d - dup_every]).astype(np.float32)
        nudged = np.where(which == 0, base, np.where(which == 1, np.nextafter(base, np.float32(np.inf)),
                                                      np.nextafter(base, np.float32(-np.inf)))).astype(np.float32)
        lw[:, d] = nudged.astype(np.float64) / -2.0      # (float)(-2 * logw) reproduces `nudged` exactly
    if big:
        lw += 3.0e5                                      # |constant| ~ 6e5: tau must scale with it
    model["log_weight"] = lw.reshape(-1)
    return model


@pytest.mark.parametrize("n_dens,dup_every,big", [(96, 7, False), (200, 3, False), (130, 5, True)])
def test_tied_screen_ties_and_ulp_neighbours(ctx, n_dens, dup_every, big):
    """the f32-screened tied kernel applies the reference's sequential f64 rule to the survivors only; ties, one-ulp
    neighbours, density counts that are not a multiple of the 64-wide chunk and huge constants must not change a bit"""
    model = _tied_adversarial(41 + n_dens, 150, n_dens, 24, dup_every, big)
    x = feats(100, 24, 43)
    assert_exact(ctx, model, x)
    import rasr_amd
    _, best = rasr_amd.GmmFeatureScorer(ctx, model).score(x)
    assert (best % dup_every != 0).any() and (best >= dup_every).any()  # duplicates do win sometimes


def test_tied_screen_non_finite_and_constant_features(ctx):
    model = synth.gmm_tied(70, 64, 16, seed=9, pooled=True)
    x = feats(64, 16, 10)
    x[3] = 0.0
    x[5, 2] = np.inf          # distance +inf for every density: no density ever beats FLT_MAX
    x[6, 0] = np.nan
    x[7] = 1e18               # squares overflow to +inf
    x[8] = 3e3                # large but finite distances (~1e8): tau scales with |min|
    assert_exact(ctx, model, x)


def _cart_adversarial(seed, n_mix, dim, pooled):
    """private-density model with exact duplicate densities and one-ulp weight neighbours inside mixtures"""
    model = synth.gmm_cart(n_mix, 1, 16, dim, seed=seed, pooled=pooled)
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    off = model["mix_offsets"]
    means, lw = model["means"], model["log_weight"]
    var = model["variances"]
    for m in range(n_mix):
        k0, k1 = int(off[m]), int(off[m + 1])
        if k1 - k0 < 3:
            continue
        a, b = k0, k1 - 1                              # first and last density of the mixture become twins
        means[model["dens_mean"][model["dens_index"][b]]] = means[model["dens_mean"][model["dens_index"][a]]]
        if not pooled:
            var[model["dens_cov"][model["dens_index"][b]]] = var[model["dens_cov"][model["dens_index"][a]]]
        base = np.float32(-2 * lw[a])
        pick = rng.integers(0, 3)
        nudged = base if pick == 0 else np.nextafter(base, np.float32(np.inf if pick == 1 else -np.inf))
        lw[b] = np.float64(nudged) / -2.0
    return model


@pytest.mark.parametrize("dim,pooled", [(40, True), (40, False), (64, True), (33, False), (16, True)])
def test_mfma_screen_twins_and_ulp_neighbours(ctx, dim, pooled):
    """MFMA-screened private-density path (one and two K-tiles, persistent and plain screen kernels): twin densities with
    equal or one-ulp-apart constants inside a mixture must resolve exactly like the reference's sequential f64 rule;
    45 mixtures (not a multiple of 16) and 700 frames (not a multiple of 256) exercise the padding"""
    model = _cart_adversarial(70 + dim, 45, dim, pooled)
    x = feats(700, dim, 71)
    assert_exact(ctx, model, x)
    import rasr_amd
    _, best = rasr_amd.GmmFeatureScorer(ctx, model).score(x)
    assert (best > 0).any()


def test_mfma_screen_operand_range(ctx):
    """features that do not fit the f16 screen operand (|x / sigma| > 65504, inf, NaN) keep every density and are evaluated
    exactly; large but representable features widen tau; tiny variances make the MODEL unscreenable (exact fallback)"""
    import rasr_amd
    model = synth.gmm_cart(30, 2, 16, 24, seed=81, pooled=True)
    x = feats(300, 24, 82)
    x[5] *= 300.0
    x[6] *= 3.0e4
    x[7] = 1.0e6
    x[8, 3] = np.inf
    x[9, 0] = np.nan
    x[10] = 0.0
    x[11] = -65000.0
    assert_exact(ctx, model, x)
    tiny = dict(model)
    tiny["variances"] = (model["variances"] * 1e-12).astype(np.float32)   # 1/sigma = 1e6: the means operand overflows f16
    assert_exact(ctx, tiny, feats(40, 24, 83))
    big = dict(model)
    big["means"] = (model["means"] * 1.0e4).astype(np.float32)
    assert_exact(ctx, big, feats(40, 24, 84) * 1.0e4)


def test_mfma_screen_many_frames_and_chunks(ctx):
    """more than one 16384-frame chunk of the screen workspace, frame count not a multiple of anything"""
    model = synth.gmm_cart(20, 1, 16, 40, seed=91, pooled=True)
    x = feats(16384 + 301, 40, 92)
    import rasr_amd
    from oracle import OracleGmm
    sc, best = rasr_amd.GmmFeatureScorer(ctx, model).score(x)
    sel = np.r_[0:64, 16300:16500, len(x) - 64:len(x)]
    osc, obest = OracleGmm(model).score(x[sel], mode=0)
    assert np.array_equal(sc[sel].view(np.uint32), osc.view(np.uint32)) and np.array_equal(best[sel], obest)


@pytest.mark.parametrize("kind", ["cart", "cart-wide", "tied"])
def test_score_stats_dev(ctx, kind):
    """amx_gmm_score_stats_dev: scores / best densities identical to amx_gmm_score_dev, best state = first arg-min over the
    states, counts and score sum accumulate -- on the screened path (arg-min fused into the exact stage), on a model the
    screen does not take (more than 16 densities per mixture) and on the tied path"""
    import torch

    import rasr_amd
    if kind == "cart":
        model = synth.gmm_cart(333, 1, 16, 40, seed=101, pooled=True)
    elif kind == "cart-wide":
        model = synth.gmm_cart(40, 10, 24, 40, seed=102, pooled=False)
    else:
        model = synth.gmm_tied(120, 64, 40, seed=103)
    T, M = 1000, len(model["mix_offsets"]) - 1
    x = feats(T, 40, 104)
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    ref_scores, ref_best = sc.score(x)
    xd = torch.from_numpy(x).cuda()
    scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
    bestd = torch.empty((T, M), dtype=torch.int32, device="cuda")
    state = torch.empty((T,), dtype=torch.int32, device="cuda")
    counts = torch.zeros((M,), dtype=torch.int64, device="cuda")
    ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
    ctx.use_torch_stream()
    for _ in range(2):
        sc.score_stats_dev(xd, T, scores, bestd, state, counts, ssum)
    torch.cuda.synchronize()
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), ref_scores.view(np.uint32))
    assert np.array_equal(bestd.cpu().numpy().astype(np.uint32), ref_best)
    want_state = ref_scores.argmin(axis=1)
    assert np.array_equal(state.cpu().numpy(), want_state)
    assert np.array_equal(counts.cpu().numpy(), 2 * np.bincount(want_state, minlength=M))
    want_sum = 2 * ref_scores.min(axis=1).astype(np.float64).sum()
    assert abs(float(ssum.item()) - want_sum) <= 1e-9 * abs(want_sum)


@pytest.mark.parametrize("kind,n_mix,T,tuning", [("cart", 333, 1000, ""), ("cart", 10000, 4200, ""), ("cart", 45, 257, ""), ("cart", 17, 31, ""),
                                                 ("cart", 1001, 700, ""), ("cart", 70, 300, "fused_waves=13"), ("cart", 70, 5000, "fused_waves=16"),
                                                 ("cart", 70, 300, "fused=0"), ("cart-wide", 40, 1000, ""), ("tied", 120, 1000, ""),
                                                 ("adversarial", 83, 515, ""), ("nan", 64, 200, "")])
def test_score_stats_u8_dev(ctx, kind, n_mix, T, tuning):
    """amx_gmm_score_stats_u8_dev: the best-density matrix as one byte per (frame, mixture).  gmm_fused_kernel<., 2, 8 | 12> writes
    it directly (8-byte pieces, n_mix % 8 != 0 -> scalar path; both wave counts); every other path (specialised / 16-wave kernels,
    two-kernel screen, per-density covariances, tied models) narrows a u32 workspace.  Scores, statistics and indices equal the u32
    form's -- which the other tests pin against the oracle -- with 0xff where it writes 0xffffffff; and Viterbi statistics
    accumulated from the byte matrix equal those from the u32 matrix bit for bit"""
    import torch

    import rasr_amd
    dim = 40
    if kind == "cart":
        model = synth.gmm_cart(n_mix, 1, 16, dim, seed=700 + n_mix, pooled=True)
    elif kind == "cart-wide":
        model = synth.gmm_cart(n_mix, 10, 24, dim, seed=702, pooled=False)
    elif kind == "tied":
        model = synth.gmm_tied(n_mix, 64, dim, seed=703)
    elif kind == "adversarial":
        model = _cart_adversarial(704, n_mix, dim, True)
    else:
        model = synth.gmm_cart(n_mix, 16, 16, dim, seed=705, pooled=True)
    x = feats(T, dim, 706 + T)
    if kind == "nan":
        x[3, 5] = np.nan
        x[150] = np.nan
    M = len(model["mix_offsets"]) - 1
    sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning=tuning)
    xd = torch.from_numpy(x).cuda()
    ctx.use_torch_stream()

    def run(dtype):
        scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
        bestd = torch.full((T, M), 7, dtype=dtype, device="cuda")
        state = torch.empty((T,), dtype=torch.int32, device="cuda")
        counts = torch.zeros((M,), dtype=torch.int64, device="cuda")
        ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
        sc.score_stats_dev(xd, T, scores, bestd, state, counts, ssum)
        torch.cuda.synchronize()
        return scores, bestd, state, counts, ssum

    s32, b32, st32, c32, sum32 = run(torch.int32)
    s8, b8, st8, c8, sum8 = run(torch.uint8)
    assert np.array_equal(s8.cpu().numpy().view(np.uint32), s32.cpu().numpy().view(np.uint32))
    want = b32.cpu().numpy().astype(np.uint32)
    assert want.max() == 0xffffffff or want.max() < 255
    assert np.array_equal(b8.cpu().numpy(), np.where(want == 0xffffffff, 255, want).astype(np.uint8))
    assert np.array_equal(st8.cpu().numpy(), st32.cpu().numpy()) and np.array_equal(c8.cpu().numpy(), c32.cpu().numpy())
    assert sum8.cpu().numpy().tobytes() == sum32.cpu().numpy().tobytes()
    if kind != "nan":  # (the u32 form indexes with 0xffffffff there too: statistics of a NaN frame are the caller's to avoid)
        acc32 = torch.zeros((sc.accumulator_size(),), dtype=torch.float64, device="cuda")
        acc8 = torch.zeros_like(acc32)
        sc.accumulate_dev(xd, T, st32, b32, M, acc32)
        sc.accumulate_dev(xd, T, st8, b8, M, acc8)
        torch.cuda.synchronize()
        a32, a8 = acc32.cpu().numpy(), acc8.cpu().numpy()
        nk = int(model["mix_offsets"][-1])
        assert a32[:nk].sum() == T and np.array_equal(a32[:nk], a8[:nk])      # density weights: whole numbers, exact
        assert np.allclose(a32, a8, rtol=1e-12, atol=0)                       # f64 sums through atomics: order of addition varies


def test_score_stats_u8_rejects_wide_mixtures(ctx):
    """a mixture of more than 255 densities cannot be indexed by a byte: AMX_ERR_UNSUPPORTED, nothing written"""
    import torch

    import rasr_amd
    model = synth.gmm_cart(3, 300, 300, 16, seed=710, pooled=True)
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    T, M = 8, 3
    xd = torch.from_numpy(feats(T, 16, 711)).cuda()
    ctx.use_torch_stream()
    with pytest.raises(rasr_amd.AmxError) as e:
        sc.score_stats_dev(xd, T, torch.empty((T, M), dtype=torch.float32, device="cuda"), torch.empty((T, M), dtype=torch.uint8, device="cuda"),
                           torch.empty((T,), dtype=torch.int32, device="cuda"), torch.zeros((M,), dtype=torch.int64, device="cuda"),
                           torch.zeros((1,), dtype=torch.float64, device="cuda"))
    assert "255" in str(e.value)


@pytest.mark.parametrize("kind", ["cart", "cart-wide", "cart-dim33", "tied", "adversarial"])
def test_best_density_of_the_aligned_mixture(ctx, kind):
    """amx_gmm_best_density_dev = AssigningContextScorer::bestDensity(e) for one mixture per frame: index and score equal entry
    (t, mixture[t]) of the full pass (which the other tests pin against the oracle) bit for bit -- also for out-of-range mixture
    indices (0xffffffff / f32 max) -- and the Viterbi statistics accumulated from it (best_density_ld = 0) equal those taken from
    the full best-density matrix"""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    dim = 33 if kind == "cart-dim33" else 40
    if kind in ("cart", "cart-dim33"):
        model = synth.gmm_cart(333, 1, 16, dim, seed=720, pooled=True)
    elif kind == "cart-wide":
        model = synth.gmm_cart(40, 10, 24, dim, seed=721, pooled=False)
    elif kind == "tied":
        model = synth.gmm_tied(120, 64, dim, seed=722)
    else:
        model = _cart_adversarial(723, 83, dim, True)
    M = len(model["mix_offsets"]) - 1
    T = 1500
    x = feats(T, dim, 724)
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    ref_scores, ref_best = sc.score(x)
    rng = np.random.Generator(np.random.PCG64(725))
    mix = rng.integers(0, M, T).astype(np.int32)
    mix[:50] = ref_scores[:50].argmin(axis=1)
    xd, md = torch.from_numpy(x).cuda(), torch.from_numpy(mix).cuda()
    bd = torch.full((T,), 7, dtype=torch.int32, device="cuda")
    sd = torch.zeros((T,), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    sc.best_density_dev(xd, T, md, bd, sd)
    torch.cuda.synchronize()
    assert np.array_equal(bd.cpu().numpy().astype(np.uint32), ref_best[np.arange(T), mix])
    assert np.array_equal(sd.cpu().numpy().view(np.uint32), ref_scores[np.arange(T), mix].view(np.uint32))
    osc, obest = OracleGmm(model).score(x[:64], mode=0)
    assert np.array_equal(bd.cpu().numpy()[:64].astype(np.uint32), obest[np.arange(64), mix[:64]])
    # statistics: per-frame indices (ld = 0) against the full matrix (ld = M)
    full = torch.from_numpy(ref_best.astype(np.int32)).cuda()
    acc_a = torch.zeros((sc.accumulator_size(),), dtype=torch.float64, device="cuda")
    acc_b = torch.zeros_like(acc_a)
    sc.accumulate_dev(xd, T, md, bd, 0, acc_a)
    sc.accumulate_dev(xd, T, md, full, M, acc_b)
    torch.cuda.synchronize()
    a, b = acc_a.cpu().numpy(), acc_b.cpu().numpy()
    nk = int(model["mix_offsets"][-1])
    assert a[:nk].sum() == T and np.array_equal(a[:nk], b[:nk]) and np.allclose(a, b, rtol=1e-12, atol=0)
    # a mixture index outside the model
    md2 = md.clone()
    md2[3] = M
    sc.best_density_dev(xd, T, md2, bd, sd)
    torch.cuda.synchronize()
    assert int(bd[3].item()) == -1 and float(sd[3].item()) == np.float32(0.5) * np.finfo(np.float32).max
    sc.best_density_dev(xd, T, md2, bd)   # scores are optional
    torch.cuda.synchronize()


def test_workspaces_regrow_between_calls(ctx):
    """the screen workspace, the fused-statistics partials and the host staging buffers of one handle grow with the batch:
    alternate small and large batches through the device and the host entry points"""
    import torch

    import rasr_amd
    model = synth.gmm_cart(100, 1, 16, 40, seed=111, pooled=True)
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    M = 100
    ctx.use_torch_stream()
    counts = torch.zeros((M,), dtype=torch.int64, device="cuda")
    ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
    total = np.zeros(M, np.int64)
    for i, T in enumerate((300, 5000, 64, 9000, 300)):
        x = feats(T, 40, 112 + i)
        want, want_best = sc.score(x)                                    # host path (staging buffers regrow)
        xd = torch.from_numpy(x).cuda()
        scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
        bestd = torch.empty((T, M), dtype=torch.int32, device="cuda")
        state = torch.empty((T,), dtype=torch.int32, device="cuda")
        sc.score_stats_dev(xd, T, scores, bestd, state, counts, ssum)    # device path (screen workspace + partials regrow)
        torch.cuda.synchronize()
        assert np.array_equal(scores.cpu().numpy().view(np.uint32), want.view(np.uint32)), T
        assert np.array_equal(bestd.cpu().numpy().astype(np.uint32), want_best), T
        assert np.array_equal(state.cpu().numpy(), want.argmin(axis=1)), T
        total += np.bincount(want.argmin(axis=1), minlength=M)
        assert np.array_equal(counts.cpu().numpy(), total), T
    from oracle import OracleGmm
    osc, _ = OracleGmm(model).score(x[:50], mode=0)
    assert np.array_equal(want[:50].view(np.uint32), osc.view(np.uint32))


def _alignment(T, n_mix, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    runs = rng.integers(0, n_mix, T // 7 + 1)
    return np.repeat(runs, 7)[:T].astype(np.uint32)       # bursty, like an aligned utterance


@pytest.mark.parametrize("pooled", [True, False])
def test_weighted_viterbi_accumulators(ctx, pooled):
    """accumulate(mixture, x, weight) with viterbi_: the best density takes the frame weight; f64 sums of w*x and (w*x)*x equal the
    oracle's sequential accumulation up to summation order (1e-12); NULL weights reproduce the unweighted statistics."""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(40, 1, 6, 40, seed=70, pooled=pooled)
    T = 2500
    x = feats(T, 40, 71)
    w = np.random.Generator(np.random.PCG64(72)).uniform(0.0, 2.0, T)
    w[::17] = 0.0
    sc, o = rasr_amd.GmmFeatureScorer(ctx, model), OracleGmm(model)
    mix = _alignment(T, 40, 73)
    osc, obest = o.score(x)
    chosen = obest[np.arange(T), mix].astype(np.uint32)
    ctx.use_torch_stream()
    xd, md, cd, wd = (torch.from_numpy(a).cuda() for a in (x, mix.astype(np.int32), chosen.astype(np.int32), w))
    acc = torch.zeros(sc.accumulator_size(), dtype=torch.float64, device="cuda")
    sc.accumulate_weighted_dev(rasr_amd.AMX_GMM_VITERBI, xd, T, md, wd, cd, 0, acc)
    torch.cuda.synchronize()
    want = o.accumulate_weighted(0, x, mix, w, chosen)
    assert np.allclose(acc.cpu().numpy(), want, rtol=1e-12, atol=1e-9)
    acc.zero_()
    sc.accumulate_weighted_dev(rasr_amd.AMX_GMM_VITERBI, xd, T, md, None, cd, 0, acc)
    torch.cuda.synchronize()
    assert np.allclose(acc.cpu().numpy(), o.accumulate(x, mix, chosen), rtol=1e-12, atol=1e-9)
    with pytest.raises(rasr_amd.AmxError):
        sc.accumulate_weighted_dev(7, xd, T, md, None, cd, 0, acc)


@pytest.mark.parametrize("kind", ["cart-pooled", "cart-private", "tied", "odd-dim"])
def test_baum_welch_accumulators(ctx, kind):
    """Baum-Welch statistics: density posteriors exp(score(e) - s_k) of the log-add scorer times the frame weight, kept above f32
    epsilon.  The device's expf / logf differ from glibc's by ulps, so posteriors agree to ~1e-6 relative; a density whose weight
    sits at the 1.2e-7 threshold may be kept on one side only (absolute term)."""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    model = {"cart-pooled": lambda: synth.gmm_cart(30, 1, 9, 40, seed=80, pooled=True),
             "cart-private": lambda: synth.gmm_cart(30, 2, 7, 24, seed=81, pooled=False),
             "tied": lambda: synth.gmm_tied(20, 150, 16, seed=82, pooled=True, alpha=1.0),
             "odd-dim": lambda: synth.gmm_cart(12, 3, 5, 7, seed=83, pooled=False)}[kind]()
    dim, n_mix, T = int(model["dim"]), len(model["mix_offsets"]) - 1, 1500
    x = (feats(T, dim, 84) * 0.6).astype(np.float32)
    w = np.random.Generator(np.random.PCG64(85)).uniform(0.2, 1.0, T)
    mix = _alignment(T, n_mix, 86)
    sc, o = rasr_amd.GmmFeatureScorer(ctx, model), OracleGmm(model)
    ctx.use_torch_stream()
    xd, md, wd = (torch.from_numpy(a).cuda() for a in (x, mix.astype(np.int32), w))
    acc = torch.zeros(sc.accumulator_size(), dtype=torch.float64, device="cuda")
    sc.accumulate_weighted_dev(rasr_amd.AMX_GMM_BAUM_WELCH, xd, T, md, wd, None, 0, acc)
    torch.cuda.synchronize()
    got = acc.cpu().numpy()
    want = o.accumulate_weighted(1, x, mix, w)
    nk = int(model["mix_offsets"][-1])
    assert abs(got[:nk].sum() - w.sum()) < 1e-3 * w.sum()            # posteriors of a frame sum to ~1 (minus the dropped tail)
    scale = np.abs(want).max()
    assert np.allclose(got, want, rtol=2e-5, atol=2e-6 * max(scale, 1.0)), np.abs(got - want).max()
    # more than one density per frame really takes part
    assert (want[:nk] > 0).sum() > n_mix


def test_baum_welch_em_does_not_decrease_likelihood(ctx):
    """size-independent property of the whole training loop on the device path: log-add scores -> Baum-Welch statistics ->
    amx_gmm_estimate -> new scorer; the total negative log-likelihood of the aligned frames must not increase (EM)."""
    import torch

    import rasr_amd
    rng = np.random.Generator(np.random.PCG64(90))
    n_mix, K, dim, T = 6, 4, 12, 6000
    true_mu = rng.standard_normal((n_mix, K, dim)) * 2.5
    mix = _alignment(T, n_mix, 91)
    comp = rng.integers(0, K, T)
    x = (true_mu[mix, comp] + rng.standard_normal((T, dim)) * rng.uniform(0.5, 1.5, dim)).astype(np.float32)
    model = synth.gmm_cart(n_mix, K, K, dim, seed=92, pooled=False)
    ctx.use_torch_stream()
    xd, md = torch.from_numpy(x).cuda(), torch.from_numpy(mix.astype(np.int32)).cuda()
    nll = []
    for it in range(4):
        sc = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="diagonal-sum")
        scores = torch.empty((T, n_mix), dtype=torch.float32, device="cuda")
        sc.score_dev(xd, T, scores, None)
        nll.append(float(scores[torch.arange(T, device="cuda"), md.long()].double().sum()))
        acc = torch.zeros(sc.accumulator_size(), dtype=torch.float64, device="cuda")
        sc.accumulate_weighted_dev(rasr_amd.AMX_GMM_BAUM_WELCH, xd, T, md, None, None, 0, acc)
        torch.cuda.synchronize()
        model = rasr_amd.gmm_estimate(model, acc.cpu().numpy(), min_observation_weight=0.0)
        assert len(model["mix_offsets"]) == n_mix + 1
    assert all(b <= a + 1e-3 * abs(a) for a, b in zip(nll, nll[1:])), nll
    assert nll[-1] < nll[0] - 0.05 * abs(nll[0]), nll


def test_split_trained_model_scores_bit_exact(ctx):
    """a trained-SHAPED model -- grown 1 -> 16 densities per state by the repository's own loop (accumulate, amx_gmm_estimate + split,
    Viterbi re-estimation: tests/trained_gmm.py) on clustered features, so the densities of a mixture are close relatives and far
    more of them pass the f16 screen than of a random-init model's -- is scored bit for bit like the oracle, scores and best
    densities, by the fused kernel, the two-kernel screen path and the evaluate-everything kernel; the loop itself behaves like
    training (every round doubles the densities, the mean score of the aligned state falls)"""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    from tests.trained_gmm import split_trained_gmm
    n_mix = 160
    model, x, align, hist = split_trained_gmm(ctx, n_mix=n_mix, dim=40, frames_per_state=500, rounds=4, iters=2, seed=5)
    ks = np.diff(model["mix_offsets"])
    assert ks.max() == 16 and ks.min() >= 4 and model["variances"].shape[0] == 1, (ks.min(), ks.max())
    dens = [h["densities"] for h in hist]        # a mean splits once it has seen 20 frames: nearly a doubling per round
    assert dens[0] == 2 * n_mix and all(1.5 * a < b <= 2 * a for a, b in zip(dens, dens[1:])), dens
    assert all(b["mean_score"] < a["mean_score"] for a, b in zip(hist, hist[1:])), hist
    T = 700
    xs = x[:T].contiguous()
    xh = xs.cpu().numpy()
    want, wbest = OracleGmm(model).score(xh, mode=0)
    fused = rasr_amd.GmmFeatureScorer(ctx, model)
    fused.screen_counts(True)
    for tuning in (None, "fused=0", "screen=0"):
        sc = fused if tuning is None else rasr_amd.GmmFeatureScorer(ctx, model, tuning=tuning)
        got, best = sc.score(xh)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (tuning, np.abs(got - want).max())
        assert np.array_equal(best, wbest), tuning
    surv, pairs = fused.screen_counts(False)
    assert pairs > 0 and surv / pairs >= 1.0, (surv, pairs)
    # the frames of a state are scored best by that state most of the time: the model has learnt the clusters
    assert (got.argmin(axis=1) == align[:T].cpu().numpy()).mean() > 0.8
    # the model right after the last split -- exact twins mean +- eps sqrt(var), the screen cannot separate them -- is exact as well
    twins = split_trained_gmm.fresh_split
    want, wbest = OracleGmm(twins).score(xh[:200], mode=0)
    sc = rasr_amd.GmmFeatureScorer(ctx, twins)
    sc.screen_counts(True)
    got, best = sc.score(xh[:200])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(best, wbest)
    surv2, pairs2 = sc.screen_counts(False)
    assert surv2 / pairs2 > 1.8, (surv2, pairs2)                       # both twins of the winner survive


def assert_simd_exact(ctx, model, x, expect_scaling=True, tuning=None):
    import rasr_amd
    from oracle import OracleGmm
    sc = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="SIMD-diagonal-maximum", tuning=tuning)
    got, best = sc.score(x)
    want, obest, scaling = OracleGmm(model).score_simd(x)
    if expect_scaling:
        assert np.float32(sc.simd_scaling()) == np.float32(scaling)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    assert np.array_equal(best, obest)
    return got


@pytest.mark.parametrize("n_mix,dim,T", [(256, 40, 300), (37, 40, 65), (50, 16, 256), (16, 64, 1), (130, 39, 700), (10, 33, 513)])
def test_simd_scorer_pooled_mfma_exact(ctx, n_mix, dim, T):
    """SIMD-diagonal-maximum on the i8 MFMA path (pooled covariance, <= 16 densities per mixture): integer scores and the first
    minimum are exact, so scores and best densities equal the oracle bit for bit; mixture counts that are not multiples of 16 / 4
    and frame counts that are not multiples of 256 exercise the guarded stores."""
    model = synth.gmm_cart(n_mix, 1, 16, dim, seed=100 + n_mix, pooled=True)
    got = assert_simd_exact(ctx, model, feats(T, dim, 101))
    # the quantised scorer approximates the float one: same best density for most frames, scores within the quantisation error
    import rasr_amd
    ref, _ = rasr_amd.GmmFeatureScorer(ctx, model).score(feats(T, dim, 101))
    assert np.median(np.abs(got - ref)) < 1.0


@pytest.mark.parametrize("kind", ["private-cov", "tied", "long-mixtures", "dim-80"])
def test_simd_scorer_general_path_exact(ctx, kind):
    """per-density covariances (one quantised feature vector per covariance), tied mixtures, > 16 densities per mixture and
    dim > 64 take the two-stage integer path"""
    model = {"private-cov": lambda: synth.gmm_cart(40, 1, 6, 24, seed=110, pooled=False),
             "tied": lambda: synth.gmm_tied(60, 96, 40, seed=111, pooled=True, alpha=1.0),
             "long-mixtures": lambda: synth.gmm_cart(12, 17, 40, 40, seed=112, pooled=True),
             "dim-80": lambda: synth.gmm_cart(20, 1, 8, 80, seed=113, pooled=True)}[kind]()
    assert_simd_exact(ctx, model, feats(333, int(model["dim"]), 114))


def test_simd_scorer_paths_agree_and_edge_values(ctx, monkeypatch):
    """the MFMA path and the general path give identical results; features far outside the quantiser's range, infinities and NaN
    follow the reference's x86 behaviour ((int)round(v) -> INT_MIN, +128 wraps, clipped to 0 / 255); an empty mixture keeps the
    initial INT_MAX score and no density; ties between densities go to the first one"""
    import rasr_amd
    model = synth.gmm_cart(45, 1, 16, 40, seed=120, pooled=True)
    off = model["mix_offsets"]
    model["means"][off[3] + 1] = model["means"][off[3]]           # twin densities with equal weights: first wins
    model["log_weight"][off[3] + 1] = model["log_weight"][off[3]]
    x = feats(200, 40, 121)
    x[5, 3] = 1e10
    x[6, 7] = -1e10
    x[7, 0] = np.inf
    x[8, 1] = -np.inf
    x[9, 2] = np.nan
    x[10] = 3e9
    x[11] *= 50
    a = assert_simd_exact(ctx, model, x)
    b = assert_simd_exact(ctx, model, x, tuning="simd_mfma=0")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # an empty mixture in the middle of the set
    ks = np.diff(off).astype(np.int64)
    ks2 = np.insert(ks, 7, 0)
    model2 = dict(model)
    model2["mix_offsets"] = np.concatenate([[0], np.cumsum(ks2)]).astype(np.uint32)
    got = assert_simd_exact(ctx, model2, x[:70])
    sc = rasr_amd.GmmFeatureScorer(ctx, model2, feature_scorer_type="SIMD-diagonal-maximum")
    _, best = sc.score(x[:70])
    assert np.all(best[:, 7] == 0xffffffff) and np.all(got[:, 7] > 1e5)


def test_simd_scorer_large_constants_fall_back(ctx):
    """weights so small that constant << 4 would not fit the packed key: the model must still score exactly (general path)"""
    model = synth.gmm_cart(20, 2, 8, 40, seed=130, pooled=True)
    model["log_weight"] = model["log_weight"] - 4.0e5
    assert_simd_exact(ctx, model, feats(100, 40, 131))


def test_simd_scorer_device_entry_and_chunks(ctx):
    """device-resident call over more frames than one internal pass handles at the full BASELINE width is covered by the bench;
    here: results of score_dev equal the host-buffer call, with and without best-density output"""
    import torch

    import rasr_amd
    model = synth.gmm_cart(100, 16, 16, 40, seed=140, pooled=True)
    sc = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="SIMD-diagonal-maximum")
    x = feats(1000, 40, 141)
    want, wbest = sc.score(x)
    ctx.use_torch_stream()
    xd = torch.from_numpy(x).cuda()
    s1 = torch.empty((1000, 100), dtype=torch.float32, device="cuda")
    b1 = torch.empty((1000, 100), dtype=torch.int32, device="cuda")
    sc.score_dev(xd, 1000, s1, b1)
    s2 = torch.empty_like(s1)
    sc.score_dev(xd, 1000, s2, None)
    torch.cuda.synchronize()
    assert np.array_equal(s1.cpu().numpy(), want) and np.array_equal(s2.cpu().numpy(), want)
    assert np.array_equal(b1.cpu().numpy().astype(np.uint32), wbest)


@pytest.mark.parametrize("kind", ["cart", "long-mixtures", "tied"])
def test_batch_int_scorer_exact(ctx, kind, monkeypatch):
    """batch-diagonal-maximum-int / -fast: the SIMD scorer's quantisation with its own constants ((s32) of an f64 difference) and an
    f32 division at the end; bit-exact on the i8 MFMA path and on the general path; pooled covariance only; no density output"""
    import rasr_amd
    from oracle import OracleGmm
    model = {"cart": lambda: synth.gmm_cart(70, 1, 16, 40, seed=150, pooled=True),
             "long-mixtures": lambda: synth.gmm_cart(9, 20, 33, 24, seed=151, pooled=True),
             "tied": lambda: synth.gmm_tied(40, 64, 16, seed=152, pooled=True, alpha=1.0)}[kind]()
    x = feats(300, int(model["dim"]), 153)
    x[3] *= 40
    want = OracleGmm(model).score_batch_int(x)
    for name in ("batch-diagonal-maximum-int", "batch-diagonal-maximum-fast"):
        got = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type=name).score(x, want_best=False)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    got = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="batch-diagonal-maximum-int", tuning="simd_mfma=0").score(x, want_best=False)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    sc = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="batch-diagonal-maximum-int")
    with pytest.raises(rasr_amd.AmxError, match="does not assign densities"):
        sc.score(x)
    private = synth.gmm_cart(10, 1, 4, 16, seed=154, pooled=False)
    with pytest.raises(rasr_amd.AmxError, match="globally pooled"):
        rasr_amd.GmmFeatureScorer(ctx, private, feature_scorer_type="batch-diagonal-maximum-int").score(feats(4, 16, 1), want_best=False)


@pytest.mark.parametrize("tuning", [None, "graph=1", "graph=1,fused_pack=0", "graph=1,fused=0"])
def test_small_batch_graph_replay_and_workspace_growth(ctx, tuning):
    """tuning graph=1: passes of <= 4096 frames on unchanged device buffers are replayed as HIP graphs from the third call on: results stay
    bit-identical when the buffer CONTENTS change, and a larger pass in between (which moves the workspaces the captured
    graphs point to) must not leave stale graphs behind.  (The default pass is ONE launch since round 6 and is not recorded at
    all; fused_pack=0 -- pack kernel + fused kernel -- and fused=0 -- three kernels -- are.)"""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(64, 1, 16, 40, seed=160, pooled=True)
    sc, o = rasr_amd.GmmFeatureScorer(ctx, model, tuning=tuning), OracleGmm(model)
    ctx.use_torch_stream()
    xd = torch.empty((256, 40), dtype=torch.float32, device="cuda")
    scores = torch.empty((256, 64), dtype=torch.float32, device="cuda")
    best = torch.empty((256, 64), dtype=torch.int32, device="cuda")
    big = torch.from_numpy(feats(6000, 40, 161)).cuda()
    big_s = torch.empty((6000, 64), dtype=torch.float32, device="cuda")
    for it in range(7):
        x = feats(256, 40, 170 + it)
        xd.copy_(torch.from_numpy(x))
        sc.score_dev(xd, 256, scores, best)
        torch.cuda.synchronize()
        want, wbest = o.score(x)
        assert np.array_equal(scores.cpu().numpy().view(np.uint32), want.view(np.uint32)), it
        assert np.array_equal(best.cpu().numpy().astype(np.uint32), wbest)
        if it == 3:                       # after the graph exists: a pass that regrows the workspaces
            sc.score_dev(big, 6000, big_s, None)
            torch.cuda.synchronize()
            assert np.array_equal(big_s.cpu().numpy().view(np.uint32), o.score(big.cpu().numpy())[0].view(np.uint32))


def test_full_size_shard_properties(ctx, monkeypatch):
    """BASELINE config 5 shard scale (10 000 states x 16 densities, 70 000 frames: more than one internal pass of 65 536):
    size-independent properties instead of a full oracle run --
      * the MFMA-screened scorer and the evaluate-everything kernel (tuning screen=0), two different algorithms, agree bit for bit
        on every score and every best-density index;
      * rows are independent: scoring the frames in another order permutes the results;
      * a sample of frames on both sides of the pass boundary equals the oracle;
      * the SIMD-diagonal-maximum scorer's i8 MFMA path and its general integer path agree bit for bit."""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(10000, 16, 16, 40, seed=5, pooled=True)
    T = 70000
    x = feats(T, 40, 200)
    ctx.use_torch_stream()
    xd = torch.from_numpy(x).cuda()

    def run(kind="diagonal-maximum", frames=xd, want_best=True, tuning=None):
        sc = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type=kind, tuning=tuning)
        s = torch.empty((frames.shape[0], 10000), dtype=torch.float32, device="cuda")
        b = torch.empty((frames.shape[0], 10000), dtype=torch.int32, device="cuda") if want_best else None
        sc.score_dev(frames, frames.shape[0], s, b)
        torch.cuda.synchronize()
        return s, b

    s_scr, b_scr = run()
    s_dir, b_dir = run(tuning="screen=0")
    assert torch.equal(s_scr.view(torch.int32), s_dir.view(torch.int32)) and torch.equal(b_scr, b_dir)
    del s_dir, b_dir
    perm = torch.randperm(T, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    s_p, b_p = run(frames=xd[perm].contiguous())
    assert torch.equal(s_p.view(torch.int32), s_scr[perm].view(torch.int32)) and torch.equal(b_p, b_scr[perm])
    del s_p, b_p
    sample = np.array([0, 1, 255, 256, 65535, 65536, 65537, T - 1])
    osc, obest = OracleGmm(model).score(x[sample])
    assert np.array_equal(s_scr[sample].cpu().numpy().view(np.uint32), osc.view(np.uint32))
    assert np.array_equal(b_scr[sample].cpu().numpy().astype(np.uint32), obest)
    del s_scr, b_scr
    s_m, b_m = run("SIMD-diagonal-maximum")
    s_g, b_g = run("SIMD-diagonal-maximum", frames=xd[:8192], tuning="simd_mfma=0")
    assert torch.equal(s_m[:8192].view(torch.int32), s_g.view(torch.int32)) and torch.equal(b_m[:8192], b_g)


def test_full_size_tied_properties(ctx, monkeypatch):
    """BASELINE config 2 at full size (4096 shared densities, 10 000 tied states, batch 256): the f32-screened (min,+) tile kernel
    and the plain f64 kernel (tuning screen=0) agree bit for bit on all 2.56 M scores and density indices; two frames equal
    the oracle; frame permutations permute the results."""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_tied(10000, 4096, 40, seed=5, pooled=True)
    x = feats(256, 40, 210)
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    a, ab = sc.score(x)
    b, bb = rasr_amd.GmmFeatureScorer(ctx, model, tuning="screen=0").score(x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(ab, bb)
    perm = np.random.Generator(np.random.PCG64(3)).permutation(256)
    c, cb = sc.score(x[perm])
    assert np.array_equal(c.view(np.uint32), a[perm].view(np.uint32)) and np.array_equal(cb, ab[perm])
    osc, obest = OracleGmm(model).score(x[[0, 255]])
    assert np.array_equal(a[[0, 255]].view(np.uint32), osc.view(np.uint32)) and np.array_equal(ab[[0, 255]], obest)


@pytest.mark.parametrize("variant", ["fused", "rows", "persist", "simple"])
@pytest.mark.parametrize("dim,pooled", [(16, True), (16, False), (24, True), (24, False), (32, True), (32, False), (33, True), (39, True),
                                        (40, True), (40, False), (45, True), (48, True), (48, False), (64, True), (64, False)])
def test_every_screened_dimension_and_operand_width(ctx, monkeypatch, dim, pooled, variant):
    """every dimension the MFMA screen is instantiated for, with pooled and per-density covariances: the screen operand is dim or
    2 dim columns wide plus two constant columns.  dim 48 pooled / dim 24 per-density are the cases whose constants land in the LAST
    16 K columns -- the MFMA step whose result an inline-asm v_min3 once read too early (found by tools/fuzz_gmm.py)."""
    if variant == "fused":
        if not (pooled and dim <= 40):
            pytest.skip("the fused kernel serves pooled covariances up to dim 40; other shapes take the two-kernel path")
    model = synth.gmm_cart(70, 1, 16, dim, seed=300 + dim, pooled=pooled)
    assert_exact(ctx, model, feats(300, dim, 301), tuning=None if variant == "fused" else "fused=0,screen_kernel=" + variant)


@pytest.mark.parametrize("prune", ["1", "0"])
@pytest.mark.parametrize("n_mix,n_dens,T,alpha", [(150, 96, 100, 2.0), (70, 64, 64, 0.1), (333, 1000, 37, 0.1), (65, 9000, 9, 0.3),
                                                  (1000, 300, 513, 5.0)])
def test_tied_pruned_and_dense_paths_exact(ctx, monkeypatch, prune, n_mix, n_dens, T, alpha):
    """the pruned path (gmm_tied.hip: nearest-density bounds -> per-tile thresholds -> exact rule over the survivors) and the dense
    (min,+) tile kernel, each forced, against the oracle: mixture counts that leave a partial last tile, density counts below /
    above the near-list size, above the LDS staging of the selection kernel (9000) and not multiples of 64; flat weights
    (alpha 5: little to prune) and peaked ones"""
    model = synth.gmm_tied(n_mix, n_dens, 24, seed=500 + n_dens, pooled=True, alpha=alpha)
    x = feats(T, 24, 501 + T)
    x[T // 2] *= 30.0
    assert_exact(ctx, model, x, tuning="tied_prune=" + prune)


def test_tied_pruned_adversarial_and_statistics(ctx, monkeypatch):
    """ties / one-ulp neighbours / huge constants through the pruned path; duplicated densities survive together (equal distances,
    so equal bounds) and the FIRST wins.  The adaptive switch: a model where nothing can be pruned (all weights equal, all
    distances equal) sends later calls to the dense kernel; results stay exact either way"""
    import rasr_amd
    for n_dens, dup_every, big in ((96, 7, False), (200, 3, False), (130, 5, True)):
        model = _tied_adversarial(41 + n_dens, 150, n_dens, 24, dup_every, big)
        assert_exact(ctx, model, feats(100, 24, 43), tuning="tied_prune=1")
    model = synth.gmm_tied(640, 512, 16, seed=77, pooled=True)
    model["means"][:] = model["means"][0]                      # every density at the same place: equal distances
    model["log_weight"][:] = np.log(1.0 / 512)                 # and equal weights: everything is a candidate
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    from oracle import OracleGmm
    x = feats(256, 16, 78)
    osc, ob = OracleGmm(model).score(x, mode=0)
    sc.screen_counts(True)
    for _ in range(24):                                        # the statistics of the first calls arrive (the counters are published
        a, b = sc.score(x)                                     # every 8th pruned call), later calls go dense
        assert np.array_equal(a.view(np.uint32), osc.view(np.uint32)) and np.array_equal(b, ob)
    surv, triples = sc.screen_counts(True)
    per_call = 512 * 256 * (640 // 64)
    assert 0 < triples <= 9 * per_call and triples % per_call == 0, triples / per_call   # pruned passes only until the first report
    assert surv > 0.5 * triples
    # equal f64 sums: the first density wins where (float)s rounded down, the LAST where it rounded up ((double)best > s again)
    assert set(np.unique(ob)) <= {0, 511}


def test_tied_pruned_list_lengths_and_frame_passes(ctx, monkeypatch):
    """the pruned path's branches by survivor-list length and call size: (a) densities almost on top of each other -- every density
    survives every tile's test: lists of 16 .. 1000 entries, i.e. a partial 16-entry block, whole 32-entry words, exactly the 256
    the wave keeps in LDS, and the longer ones that go through the unscreened loop; (b) equal weights and equal distances: every
    list position is a candidate of every mixture (the rule walks whole lists, the first / last density wins); (c) more frames than
    one pass of the tied kernels takes (4096), with the frame count not a multiple of anything"""
    for n_dens in (16, 33, 64, 255, 256, 257, 1000):
        model = synth.gmm_tied(130, n_dens, 16, seed=900 + n_dens, pooled=True)
        model["means"] = (model["means"][:1] + np.float32(1e-4) * model["means"]).astype(np.float32)
        assert_exact(ctx, model, feats(70, 16, 901), tuning="tied_prune=1")
    for n_dens in (40, 256, 300):
        model = synth.gmm_tied(70, n_dens, 16, seed=910 + n_dens, pooled=True)
        model["means"][:] = model["means"][0]
        model["log_weight"][:] = np.log(1.0 / n_dens)
        assert_exact(ctx, model, feats(33, 16, 911), tuning="tied_prune=1")
    model = synth.gmm_tied(100, 64, 16, seed=920, pooled=True)
    assert_exact(ctx, model, feats(4096 + 777, 16, 921), tuning="tied_prune=1")


def test_tied_pruned_lists_that_are_not_the_identity(ctx, monkeypatch):
    """the pruned path's frame-major distance image: written by the distance kernel when the shared list names every density once
    (here: a permutation, list position != density id, and a list that skips densities), by the transposing kernel when a density
    is listed twice (the duplicate has the same distance; the first position wins ties)"""
    rng = np.random.Generator(np.random.PCG64(930))
    for variant in ("permutation", "subset", "duplicate"):
        n_dens, n_mix = 200, 130
        model = synth.gmm_tied(n_mix, n_dens, 16, seed=931, pooled=True)
        if variant == "permutation":
            lst = rng.permutation(n_dens).astype(np.uint32)
        elif variant == "subset":
            lst = np.sort(rng.choice(n_dens, 150, replace=False)).astype(np.uint32)
        else:
            lst = np.arange(n_dens, dtype=np.uint32)
            lst[37] = 5
            lst[150] = 5
        k = len(lst)
        model["dens_index"] = np.tile(lst, n_mix)
        model["mix_offsets"] = (np.arange(n_mix + 1, dtype=np.uint64) * k).astype(np.uint32)
        g = rng.gamma(0.1, 1.0, (n_mix, k)) + 1e-30
        model["log_weight"] = np.log(g / g.sum(axis=1, keepdims=True)).reshape(-1).astype(np.float64)
        assert_exact(ctx, model, feats(70, 16, 932), tuning="tied_prune=1")


@pytest.mark.parametrize("pooled", [True, False])
def test_tied_non_finite_frames_keep_the_initial_result(ctx, pooled):
    """a frame with an inf / NaN / 1e30 feature has no finite density score: the reference keeps (FLT_MAX / 2, no density).  On the
    uniform tied path the chunk padding rows (density count not a multiple of 64) must not be taken for candidates then
    (found by tools/fuzz_gmm.py)."""
    model = synth.gmm_tied(30, 150, 32, seed=320, pooled=pooled, alpha=1.0)
    x = feats(70, 32, 321)
    x[5, 3] = np.inf
    x[6, 0] = np.nan
    x[7, 9] = 1e30
    x[8, 1] = -np.inf
    x[9] *= np.float32(1e4)
    assert_exact(ctx, model, x)


@pytest.mark.parametrize("dim,n_dens,T", [(40, 1000, 256), (39, 257, 70), (33, 300, 1), (16, 64, 513), (45, 130, 100), (48, 96, 65),
                                          (50, 200, 40)])
@pytest.mark.parametrize("contract", ["off", "fma"])
def test_tied_list_order_distances(ctx, dim, n_dens, T, contract, monkeypatch):
    """the pruned path's distance kernel (gmm_dist_list_kernel: lane = list position, tables transposed at creation) against the
    reference arithmetic, in both contracts, for every frames-per-wave setting and against the density-major kernel it replaces
    (dist_list=0); odd dimensions take the scalar tail, dim 50 has no instance and falls back; pooled and per-density covariances"""
    import rasr_amd
    from oracle import OracleGmm
    for pooled in (True, False):
        model = synth.gmm_tied(130, n_dens, dim, seed=940 + n_dens, pooled=pooled)
        x = feats(T, dim, 941)
        if T > 8:
            x[3] *= 25.0
        osc, ob = OracleGmm(model, contract=contract).score(x, mode=0)
        for dl in ("dist_list=1", "dist_list=0", "dist_list=2", "dist_list=16", "dist_list=64", "near_fused=0"):
            sc, best = rasr_amd.GmmFeatureScorer(ctx, model, tuning="tied_prune=1,contract=%s,%s" % (contract, dl)).score(x)
            assert np.array_equal(sc.view(np.uint32), osc.view(np.uint32)), (dl, pooled, np.abs(sc - osc).max())
            assert np.array_equal(best, ob), (dl, pooled)


@pytest.mark.parametrize("n_mix", [4096, 4100, 5000, 8500])
def test_tied_bound_segments_over_the_xcds(ctx, n_mix):
    """the bound kernel gives the whole rounds of eight 512-mixture table segments a fixed home XCD and deals the left-over (segment, frame)
    units round the XCDs, the pruned kernel does the same with its tiles: mixture counts with 8 / 9 / 10 / 17 segments (0, 1, 2 and 1 left
    over) and 64 / 65 / 79 / 133 tiles (0, 1, 7 and 5 left over), a frame count that is no multiple of anything"""
    model = synth.gmm_tied(n_mix, 160, 24, seed=960 + n_mix, pooled=True)
    x = feats(45, 24, 961)
    x[11] *= 20.0
    assert_exact(ctx, model, x, tuning="tied_prune=1")


def test_tied_near_keys_survive_calls_of_any_shape(ctx):
    """the list-order distance kernel keeps the frame's near densities as atomic minima over 64-bit keys that tied_list_kernel puts
    back into their empty state: calls of different lengths on ONE scorer (more frames, fewer, more again; device buffers, so that
    repeated calls are graph replays), NaN / inf frames in between, and a second scorer created and dropped in between -- every
    result bit-exact; then the same sequence with the keys from tied_near_kernel (near_fused=0)"""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_tied(300, 700, 40, seed=955, pooled=True)
    orc = OracleGmm(model)
    ctx.use_torch_stream()
    for tuning in ("tied_prune=1,graph=1", "tied_prune=1", "tied_prune=1,near_fused=0,graph=1"):
        sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning=tuning)
        for rep, T in enumerate((256, 31, 256, 256, 256, 1, 700, 256, 256)):
            x = feats(T, 40, 956 + rep)
            if rep == 4:
                x[7, 3] = np.nan
                x[9, 0] = np.inf
            xd = torch.from_numpy(x).cuda()
            scores = torch.empty((T, 300), dtype=torch.float32, device="cuda")
            best = torch.empty((T, 300), dtype=torch.int32, device="cuda")
            for _ in range(3):                              # plain, recorded, replayed
                scores.fill_(-1.0)
                sc.score_dev(xd, T, scores, best)
            torch.cuda.synchronize()
            osc, ob = orc.score(x, mode=0)
            assert np.array_equal(scores.cpu().numpy().view(np.uint32), osc.view(np.uint32)), (tuning, rep, T)
            assert np.array_equal(best.cpu().numpy().astype(np.uint32), ob), (tuning, rep, T)
            if rep == 2:
                other = rasr_amd.GmmFeatureScorer(ctx, synth.gmm_tied(64, 128, 40, seed=957, pooled=True), tuning="tied_prune=1")
                other.score(feats(50, 40, 958))
                del other


# ---- gmm_fused_kernel (gmm_fused.hip): screen + exact evaluation in one kernel, pooled covariance, dim <= 40

def _screen_worst_case(dim, n_mix, seed):
    """Model and frames that drive the f16 screen's error towards its analytic bound.

    Variance 1, so the screen operand of a density is a = -2 mu and the frame operand is x itself.  Frame m and mixture m belong
    together: every |x_i| and every |a_i| sits 0.49 ulp(f16) off an f16 grid point just above 1.0 (the largest RELATIVE rounding
    error f16 has), all with the same magnitude (Cauchy-Schwarz, which the bound uses, is then tight).  Density A = slot sa has
    a_i = +(1 + (k_i + 0.49 t_i) 2^-10), density B = slot sb is its mirror image -A, t_i = sign(x_i), so that
      * A's operand rounds so that its estimate falls by sum 0.49 * 2^-10 |x_i|, B's so that it rises by as much,
      * the frame operand's own rounding error (-0.49 * 2^-10 per component) is multiplied by a_B - a_A ~ -2: it raises B's
        estimate relative to A's once more, by twice that amount,
    i.e. all four error terms of the difference point the same way: B looks 40 * 1.96 * 2^-10 ~ 0.077 worse than A to the screen
    (the bound tau ~ 2.2e-3 * |a| * |x| ~ 0.088 has to cover exactly this), while the mixture weights are set so that B's TRUE score
    is the smaller one by 1e-3.  A screen that drops B returns the wrong score and the wrong density."""
    rng = np.random.Generator(np.random.PCG64(seed))
    model = synth.gmm_cart(n_mix, 16, 16, dim, seed=seed, pooled=True)
    model["variances"][:] = 1.0
    model["means"] += 12.0                      # every other slot: far away
    u = 2.0 ** -10
    T = n_mix
    t = rng.choice(np.array([-1.0, 1.0]), (T, dim))
    j = rng.integers(1, 24, (T, dim))
    x = (t * (1.0 + (j + 0.49 * t) * u)).astype(np.float32)          # x_hat - x = -0.49 u for either sign of x
    lw = model["log_weight"]
    slots = []
    for m in range(n_mix):
        k = rng.integers(1, 24, dim)
        a_A = 1.0 + (k + 0.49 * t[m]) * u                            # a_hat_A - a_A = -0.49 u t
        mu_A = (-0.5 * a_A).astype(np.float32)
        mu_B = -mu_A                                                  # a_B = -a_A: a_hat_B - a_B = +0.49 u t
        sa, sb = (0, 15) if rng.integers(0, 2) else (15, 0)
        base = 16 * m
        model["means"][base + sa] = mu_A
        model["means"][base + sb] = mu_B
        xd = x[m].astype(np.float64)
        d_A = ((mu_A.astype(np.float64) - xd) ** 2).sum()
        d_B = ((mu_B.astype(np.float64) - xd) ** 2).sum()
        lw[base + sa] = -3.0
        lw[base + sb] = -3.0 + 0.5 * (d_B - d_A + 2.0e-3)            # -2 lw_B + d_B = -2 lw_A + d_A - 2e-3 (scores are halved)
        slots.append((sa, sb))
    return model, x, slots


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("dim", [40, 24])
def test_screen_threshold_worst_case_model(ctx, monkeypatch, dim, fused):
    """the constructed worst case for the f16 screen (all rounding errors aligned against the true winner) is scored bit-exactly
    by the fused kernel and by the two-kernel path, and the construction really has the disadvantaged density win"""
    import rasr_amd
    from oracle import OracleGmm
    n_mix = 96
    model, x, slots = _screen_worst_case(dim, n_mix, 900 + dim)
    # what the screen sees: f16-rounded operands, difference of the two dot products against the true difference
    a16 = (-2.0 * model["means"]).astype(np.float16).astype(np.float64)
    x16 = x.astype(np.float16).astype(np.float64)
    worst = np.inf
    for m, (sa, sb) in enumerate(slots):
        est = (a16[16 * m + sb] - a16[16 * m + sa]) @ x16[m]
        true = (-2.0 * (model["means"][16 * m + sb].astype(np.float64) - model["means"][16 * m + sa].astype(np.float64))) @ x[m].astype(np.float64)
        worst = min(worst, est - true)
    assert worst > 1.9 * 2.0 ** -10 * dim                            # every pair: all four error terms aligned, ~1.96 * 2^-10 per dimension (86 % of tau)
    osc, obest = OracleGmm(model).score(x, mode=0)
    won = sum(int(obest[m, m]) == sb for m, (sa, sb) in enumerate(slots))
    assert won == n_mix, won                                          # B is the reference's winner on its frame
    sc, best = rasr_amd.GmmFeatureScorer(ctx, model, tuning="fused=" + fused).score(x)
    assert np.array_equal(sc.view(np.uint32), osc.view(np.uint32)), np.abs(sc - osc).max()
    assert np.array_equal(best, obest)


@pytest.mark.parametrize("n_mix,T", [(1, 1), (7, 31), (16, 33), (17, 256), (45, 257), (48, 1000), (333, 700), (1000, 64)])
def test_fused_shapes_exact(ctx, n_mix, T):
    """mixture counts around the 16-mixture tile (partial last tile, n_mix % 4 != 0 -> scalar store path) and frame counts around
    the 32-frame wave / 256-frame workgroup, mixtures of 1..16 densities; scores and best densities bit-exact vs the oracle"""
    model = synth.gmm_cart(n_mix, 1, 16, 40, seed=400 + n_mix, pooled=True)
    assert_exact(ctx, model, feats(T, 40, 401 + T))


@pytest.mark.parametrize("dim", [16, 24, 32, 33, 39, 40])
@pytest.mark.parametrize("contract", ["off", "fma"])
def test_fused_kernel_packs_its_own_operand_rows(ctx, dim, contract):
    """gmm_fused_kernel packs the frames' f16 operand rows itself (round 6; fused_pack=0: the rows gmm_screen_pack_kernel wrote): both
    forms bit-exact against the oracle, with frames that do not fit f16 (every slot kept), NaN / inf frames, a frame count that leaves
    lanes behind the last frame, and equal survivor statistics within a few borderline densities (the two forms sum the row norms in
    a different order)"""
    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(83, 1, 16, dim, seed=470 + dim, pooled=True)
    x = feats(333, dim, 471)
    x[5] *= np.float32(3e5)                     # v = x / sigma beyond 65504
    x[6, 1] = np.nan
    x[7, 0] = np.inf
    x[8] = 0.0
    osc, ob = OracleGmm(model, contract=contract).score(x, mode=0)
    counts = []
    for pack in ("1", "0"):
        sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning="contract=%s,fused_pack=%s" % (contract, pack))
        sc.screen_counts(True)
        got, best = sc.score(x)
        assert np.array_equal(got.view(np.uint32), osc.view(np.uint32)), (pack, np.abs(got - osc).max())
        assert np.array_equal(best, ob), pack
        counts.append(sc.screen_counts(True))
    assert counts[0][1] == counts[1][1] and abs(counts[0][0] - counts[1][0]) <= 8, counts


@pytest.mark.parametrize("dim", [16, 24, 32, 33, 39, 40])
def test_fused_adversarial_twins(ctx, dim):
    """twin densities / one-ulp weight neighbours: the further survivors of a mixture go through the divergent loop in slot order"""
    model = _cart_adversarial(410 + dim, 83, dim, True)
    x = feats(515, dim, 411)
    assert_exact(ctx, model, x)


@pytest.mark.parametrize("n_mix,dim,T", [(70, 40, 300), (333, 40, 700), (48, 24, 1000), (17, 16, 257), (45, 33, 513), (1000, 40, 3000)])
def test_fused_specialised_waves_variant_exact(ctx, n_mix, dim, T):
    """gmm_fused_spec_kernel (tuning fused_waves=13: four screen waves that run the MFMA screen one tile ahead and issue the DMA,
    eight exact waves that evaluate the survivors; masks handed over through LDS behind three barriers per tile) -- the form the
    round-3 review asked for; slower than gmm_fused_kernel (6.0 vs 4.75 ms), kept for A/B runs: scores, best densities and the fused
    best-state statistics bit-exact against the oracle, partial last tiles and frame counts off the 256-frame workgroup included"""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(n_mix, 1, 16, dim, seed=400 + n_mix, pooled=True)
    x = feats(T, dim, 401 + T)
    x[T // 3] *= 40.0
    want, wbest = OracleGmm(model).score(x, mode=0)
    sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning="fused_waves=13")
    ctx.use_torch_stream()
    xd = torch.from_numpy(x).cuda()
    s = torch.empty((T, n_mix), dtype=torch.float32, device="cuda")
    b = torch.empty((T, n_mix), dtype=torch.int32, device="cuda")
    st = torch.empty((T,), dtype=torch.int32, device="cuda")
    cnt = torch.zeros((n_mix,), dtype=torch.int64, device="cuda")
    ss = torch.zeros((1,), dtype=torch.float64, device="cuda")
    sc.score_stats_dev(xd, T, s, b, st, cnt, ss)
    torch.cuda.synchronize()
    g = s.cpu().numpy()
    assert np.array_equal(g.view(np.uint32), want.view(np.uint32)), np.abs(g - want).max()
    assert np.array_equal(b.cpu().numpy().astype(np.uint32), wbest)
    assert np.array_equal(st.cpu().numpy(), g.argmin(axis=1))
    assert np.array_equal(cnt.cpu().numpy(), np.bincount(g.argmin(axis=1), minlength=n_mix))


def test_fused_equals_two_kernel_path_with_stats(ctx, monkeypatch):
    """the fused kernel and round 1's two kernels (tuning fused=0) agree bit for bit on scores, best densities, best states,
    counts and the score sum; frames that do not fit the f16 operand keep every slot in both; without a best-density buffer too"""
    import torch

    import rasr_amd
    model = synth.gmm_cart(1203, 1, 16, 40, seed=420, pooled=True)
    T, M = 5000, 1203
    x = feats(T, 40, 421)
    x[17] *= 3.0e4
    x[18, 5] = np.nan
    x[4000] = 1.0e6
    xd = torch.from_numpy(x).cuda()
    ctx.use_torch_stream()
    res = {}
    for mode in ("1", "0"):
        sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning="fused=" + mode)
        scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
        bestd = torch.empty((T, M), dtype=torch.int32, device="cuda")
        state = torch.empty((T,), dtype=torch.int32, device="cuda")
        counts = torch.zeros((M,), dtype=torch.int64, device="cuda")
        ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
        sc.score_stats_dev(xd, T, scores, bestd, state, counts, ssum)
        only = torch.empty((T, M), dtype=torch.float32, device="cuda")
        sc.score_dev(xd, T, only, None)
        torch.cuda.synchronize()
        res[mode] = [a.cpu().numpy() for a in (scores, bestd, state, counts, only)] + [float(ssum.item())]
        del sc
    for a, b in zip(res["1"][:5], res["0"][:5]):
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    assert np.array_equal(res["1"][0].view(np.uint32), res["1"][4].view(np.uint32))
    assert abs(res["1"][5] - res["0"][5]) <= 1e-9 * abs(res["0"][5])
    from oracle import OracleGmm
    sel = np.r_[0:40, 3990:4010]
    osc, obest = OracleGmm(model).score(x[sel], mode=0)
    assert np.array_equal(res["1"][0][sel].view(np.uint32), osc.view(np.uint32)) and np.array_equal(res["1"][1][sel].astype(np.uint32), obest)


def test_fused_small_batch_splits_the_model(ctx):
    """config 3 shape: 256 frames x 10 000 mixtures -- one frame tile, the model's 625 tiles split over the CUs; best states from
    the per-split partials"""
    import torch

    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(10000, 16, 16, 40, seed=5, pooled=True)
    T, M = 256, 10000
    x = feats(T, 40, 431)
    sc = rasr_amd.GmmFeatureScorer(ctx, model)
    xd = torch.from_numpy(x).cuda()
    scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
    bestd = torch.empty((T, M), dtype=torch.int32, device="cuda")
    state = torch.empty((T,), dtype=torch.int32, device="cuda")
    counts = torch.zeros((M,), dtype=torch.int64, device="cuda")
    ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
    ctx.use_torch_stream()
    sc.score_stats_dev(xd, T, scores, bestd, state, counts, ssum)
    torch.cuda.synchronize()
    got = scores.cpu().numpy()
    sel = np.r_[0:3, 31:34, 253:256]
    osc, obest = OracleGmm(model).score(x[sel], mode=0)
    assert np.array_equal(got[sel].view(np.uint32), osc.view(np.uint32))
    assert np.array_equal(bestd.cpu().numpy()[sel].astype(np.uint32), obest)
    assert np.array_equal(state.cpu().numpy(), got.argmin(axis=1))
    assert int(counts.sum().item()) == T


# ---- preselection-batch-float (Mm::BatchPreselectionFloatFeatureScorer + Mm::FloatDensityClustering)

@pytest.mark.parametrize("n_mix,kmax,dim,clusters,select", [(50, 8, 24, 16, 4), (300, 16, 40, 256, 32), (20, 3, 40, 256, 32), (64, 4, 33, 8, 8),
                                                            (70, 6, 8, 16, 5), (40, 5, 50, 8, 3)])   # dims 8 / 50: the runtime-dimension kernels
def test_preselection_batch_float_exact(ctx, n_mix, kmax, dim, clusters, select):
    """clustering (glibc rand() initialisation restated in the product vs libc's in the oracle, k-means assignment on the GPU vs
    the oracle's loops) and the preselected scores bit-exact against the oracle; mixtures without an active density score the
    back-off score; selecting every cluster reproduces batch-diagonal-maximum-float; fewer densities than clusters reduces the
    cluster count like the reference"""
    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(n_mix, 1, kmax, dim, seed=500 + n_mix, pooled=True)
    x = feats(333, dim, 501)
    sc = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="preselection-batch-float")
    sc.set_preselection(clusters, select, 5, 40000.0)
    got = sc.score(x, want_best=False)
    want, wcof, wcm = OracleGmm(model).score_preselection_float(x, clusters, select, 5, 40000.0)
    cof, cm = sc.preselection_clustering()
    assert cm.shape[0] == wcm.shape[0] == min(clusters, len(wcof))
    assert np.array_equal(cof, wcof)
    assert np.array_equal(cm.view(np.uint32), wcm[:, :dim].view(np.uint32))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    if select < cm.shape[0]:
        assert (got == 40000.0).any()
    sc.set_preselection(clusters, min(clusters, len(wcof)), 5, 123.0)          # all clusters active: plain batch-float scores
    full = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="batch-diagonal-maximum-float").score(x, want_best=False)
    assert np.array_equal(sc.score(x, want_best=False).view(np.uint32), full.view(np.uint32))


def test_preselection_errors(ctx):
    import rasr_amd
    sc = rasr_amd.GmmFeatureScorer(ctx, synth.gmm_cart(10, 2, 4, 40, seed=9, pooled=False), feature_scorer_type="preselection-batch-float")
    with pytest.raises(rasr_amd.AmxError, match="pooled covariance"):
        sc.score(feats(3, 40, 1), want_best=False)
    sc = rasr_amd.GmmFeatureScorer(ctx, synth.gmm_cart(10, 2, 4, 40, seed=9, pooled=True), feature_scorer_type="preselection-batch-float")
    sc.set_preselection(8, 16)
    with pytest.raises(rasr_amd.AmxError, match="select-clusters"):
        sc.score(feats(3, 40, 1), want_best=False)
    with pytest.raises(rasr_amd.AmxError):
        sc.set_preselection(300, 16)


@pytest.mark.parametrize("n_mix,kmax,dim,clusters,select", [(200, 16, 40, 64, 8), (50, 8, 24, 16, 4), (300, 12, 33, 256, 32), (3, 3, 16, 256, 2),
                                                            (120, 16, 39, 32, 32)])
def test_preselection_batch_int_exact(ctx, n_mix, kmax, dim, clusters, select):
    """preselection-batch-int: integer k-means over the quantised means (same glibc rand() seeds, s32 distances, cluster means
    truncated to u8) and the preselected integer scores bit-exact against the oracle; a mixture without an active density scores
    (f32)INT_MAX / scale; selecting every cluster reproduces batch-diagonal-maximum-int; any dimension (runtime loop)"""
    import rasr_amd
    from oracle import OracleGmm
    model = synth.gmm_cart(n_mix, 1, kmax, dim, seed=600 + n_mix, pooled=True)
    x = feats(333, dim, 601)
    x[7] *= 40.0                                   # saturates the quantiser
    sc = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="preselection-batch-int")
    sc.set_preselection(clusters, select, 5, 0.0)
    got = sc.score(x, want_best=False)
    want, wcof, wcm = OracleGmm(model).score_preselection_int(x, clusters, select, 5)
    cof, cm = sc.preselection_clustering()
    assert cm.shape[0] == wcm.shape[0] == min(clusters, len(wcof))
    assert np.array_equal(cof, wcof) and np.array_equal(cm, wcm.astype(np.float32))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    if select < cm.shape[0] and n_mix > 10:
        assert (got > 1e5).any()                   # INT_MAX / scale: some mixture had no active density
    sc.set_preselection(clusters, min(clusters, len(wcof)), 5, 0.0)
    full = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="batch-diagonal-maximum-int").score(x, want_best=False)
    assert np.array_equal(sc.score(x, want_best=False).view(np.uint32), full.view(np.uint32))


def test_preselection_batch_int_errors(ctx):
    import rasr_amd
    sc = rasr_amd.GmmFeatureScorer(ctx, synth.gmm_cart(10, 2, 4, 40, seed=9, pooled=False), feature_scorer_type="preselection-batch-int")
    with pytest.raises(rasr_amd.AmxError, match="pooled"):
        sc.score(feats(3, 40, 1), want_best=False)
    sc = rasr_amd.GmmFeatureScorer(ctx, synth.gmm_cart(10, 2, 4, 40, seed=9, pooled=True), feature_scorer_type="preselection-batch-int")
    sc.set_preselection(8, 16)
    with pytest.raises(rasr_amd.AmxError, match="select-clusters"):
        sc.score(feats(3, 40, 1), want_best=False)


@pytest.mark.parametrize("tuning", ["graph=1", None])
def test_tied_statistics_stay_consistent_when_the_host_runs_ahead(ctx, tuning):
    """200 passes enqueued back to back without a synchronisation (graph replays from the third on): the survivor statistic the host reads
    asynchronously must not mistake the device's lag for a high surviving fraction -- every pass takes the pruned path (regression: the
    denominator used to be counted on the host at submission time, the numerator arrived late, and the ratio of a later window came out
    20 times too high, which sent a perfectly prunable model to the dense kernel)"""
    import torch

    import rasr_amd
    model = synth.gmm_tied(1000, 4096, 40, seed=88, pooled=True)      # prunable: ~2 % of the (density, frame, tile) triples survive
    sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning=tuning)       # graph=1: replays from the third pass on; None: plain launches
    T, M = 256, 1000
    x = torch.from_numpy(feats(T, 40, 89)).cuda()
    scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
    best = torch.empty((T, M), dtype=torch.int32, device="cuda")
    ctx.use_torch_stream()
    sc.screen_counts(True)
    for _ in range(200):
        sc.score_dev(x, T, scores, best)
    torch.cuda.synchronize()
    surv, triples = sc.screen_counts(True)
    assert triples == 200 * 4096 * T * 16, triples           # every pass was a pruned one
    assert 0 < surv < 0.05 * triples, surv / triples
    from oracle import OracleGmm
    osc, ob = OracleGmm(model).score(x.cpu().numpy())
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), osc.view(np.uint32)) and np.array_equal(best.cpu().numpy().astype(np.uint32), ob)


@pytest.mark.parametrize("tuning", [None, "graph=1", "graph=1,fused_pack=0"])
def test_fused_survivor_statistics_under_graph_replay(ctx, tuning):
    """decoder-sized passes on unchanged buffers are replayed as a HIP graph from the third call on: the (frame, mixture) pairs the
    statistic is normalised by must count the replays too (regression: only the captured call was counted, 1.02 survivors per mixture
    were reported as 1.8), and switching the counter off and on drops the graphs recorded with the other setting"""
    import torch

    import rasr_amd
    model = synth.gmm_cart(300, 16, 16, 40, seed=71, pooled=True)
    sc = rasr_amd.GmmFeatureScorer(ctx, model, tuning=tuning)   # (None: one launch per pass, not recorded; fused_pack=0: two launches, replayed)
    T, M = 256, 300
    x = torch.from_numpy(feats(T, 40, 72)).cuda()
    scores = torch.empty((T, M), dtype=torch.float32, device="cuda")
    best = torch.empty((T, M), dtype=torch.int32, device="cuda")
    ctx.use_torch_stream()
    for _ in range(4):                       # graphs recorded without the counter
        sc.score_dev(x, T, scores, best)
    sc.screen_counts(True)
    for _ in range(9):
        sc.score_dev(x, T, scores, best)
    torch.cuda.synchronize()
    surv, pairs = sc.screen_counts(True)
    assert pairs == 9 * T * M, pairs
    assert pairs <= surv < 1.3 * pairs, surv / pairs
    one = surv // 9
    assert surv == 9 * one                   # every pass evaluates the same densities
    sc.screen_counts(False)
    for _ in range(4):
        sc.score_dev(x, T, scores, best)
    torch.cuda.synchronize()
    assert sc.screen_counts(False) == (0, 0)
