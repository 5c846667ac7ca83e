"""GPU parity of AMX_PREC_F16MX (f16 hi.hi MFMA + one MX-fp4 scaled MFMA for both cross terms, rasr_amd/csrc/ffnn_mx.hpp) through
the C ABI against the oracle: the same bars as the split-bf16 path, plus the numbers the round-3 review asked for -- arg-min
mismatches over ALL frames, the frames a gap rule excludes, the worst pure relative error."""
import json

import numpy as np
import pytest

from tests import synth
from tests.parity import nn_parity_report

pytestmark = pytest.mark.gpu


def feats(T, dim, seed):
    return np.random.Generator(np.random.PCG64(seed)).standard_normal((T, dim)).astype(np.float32)


@pytest.mark.parametrize("T", [1, 100, 129, 1024])
def test_f16mx_path_meets_the_fp32_bar(ctx, T):
    """<= 1e-4 relative (+1e-4 absolute) against f64 accumulation, arg-min state identical, ragged shapes (small-batch tiles)"""
    import rasr_amd
    from oracle import oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([440, 256, 300, 1000], seed=7)
    x = feats(T, 440, 6)
    got = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="f16mx").score(x)
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=True)
    rep = nn_parity_report(got, want)
    assert rep["bar_violations"] == 0 and rep["worst_pure_relative"] <= 1e-4, rep
    assert np.array_equal(got.argmin(axis=1), want.argmin(axis=1)), rep


@pytest.mark.parametrize("act", [1, 2, 3])
def test_f16mx_activations(ctx, act):
    import rasr_amd
    from oracle import oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([64, 130, 77], seed=17, act=act)
    x = feats(50, 64, 18)
    got = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=0.6, precision="f16mx").score(x)
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=0.6, acc64=True)
    assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-4), np.abs(got - want).max()
    # ... and WITHOUT the absolute term, relative to the frame's score scale: |d| <= 1e-4 max_e |ref(t, e)|.  (Pointwise, a score that
    # passes near zero has no bounded relative error in any finite arithmetic: on this 64-130-77 network the worst pointwise figure over
    # |ref| > 1e-2 is 6.1e-4 for f16mx, 1.6e-4 for split bf16 and 6.8e-6 for the f32 MFMA path, at absolute errors of 6e-5 / 2e-5 / 1e-6 on
    # scores up to 6 -- DESIGN.md section 6; the full-size tests assert the pointwise form, where every score carries a prior of ~10.)
    assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want).max(axis=1, keepdims=True)), (np.abs(got - want) / np.abs(want).max(axis=1, keepdims=True)).max()


def test_f16mx_single_layer_and_tiny_shapes(ctx):
    """no hidden layer; input dimension below one K-tile; one output"""
    import rasr_amd
    from oracle import oracle_ffnn_score
    for dims, T, seed in (([7, 5], 3, 1), ([33, 1], 70, 2), ([440, 2048, 1], 260, 3), ([1, 64, 300], 513, 4)):
        Ws, bs, acts, logp = synth.ffnn(dims, seed=seed)
        x = feats(T, dims[0], seed + 10)
        got = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="f16mx").score(x)
        want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=True)
        assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-4), (dims, np.abs(got - want).max())
        # ... and without the absolute term, relative to the frame's score scale (see test_f16mx_activations); a one-output network has
        # no other score in the frame, so its scale is the largest score of the batch
        scale = np.abs(want).max(axis=1, keepdims=True) if want.shape[1] > 1 else np.abs(want).max()
        assert np.all(np.abs(got - want) <= 1e-4 * scale), (dims, (np.abs(got - want) / scale).max())


def test_f16mx_config4_full_size_against_the_oracle(ctx, capsys):
    """BASELINE config 4 at full size -- 440-6x2048-10000, batch 1024 -- against the f64-accumulating oracle on EVERY score:
    |delta| <= 1e-4 |ref| + 1e-4 AND the pure relative error over |ref| > 1e-2 stays below 1e-4; the arg-min state equals the
    oracle's and the exact-f32 MFMA path's on ALL frames outside a 1e-5 gap rule (a third of the worst relative error the scheme shows), and the
    counts are printed; the fused statistics agree with a recount of the scores"""
    import torch

    import rasr_amd
    from oracle import oracle_ffnn_score
    dims = [440] + [2048] * 6 + [10000]
    Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
    T = 1024
    x = feats(T, 440, 6)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="f16mx")
    got = nn.score(x)
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=True)
    f32 = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="fp32").score(x)
    rep = nn_parity_report(got, want, other=f32, gap=1e-5)
    with capsys.disabled():
        print("\nf16mx config 4 parity:", json.dumps(rep))
    assert rep["bar_violations"] == 0 and rep["worst_over_bar"] <= 0.5, rep
    assert rep["worst_pure_relative"] <= 1e-4, rep
    assert rep["argmin_mismatches_outside_gap_rule"] == 0 and rep["frames_excluded_by_gap_rule"] <= 0.01 * T, rep
    assert rep["argmin_mismatches"] <= rep["frames_excluded_by_gap_rule"], rep
    xd = torch.from_numpy(x).cuda()
    scores = torch.empty((T, 10000), dtype=torch.float32, device="cuda")
    state = torch.empty((T,), dtype=torch.int32, device="cuda")
    counts = torch.zeros((10000,), dtype=torch.int64, device="cuda")
    ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
    ctx.use_torch_stream()
    nn.score_stats_dev(xd, 440, T, scores, state, counts, ssum)
    torch.cuda.synchronize()
    sc = scores.cpu().numpy()
    assert np.array_equal(sc.view(np.uint32), got.view(np.uint32))
    assert np.array_equal(state.cpu().numpy(), sc.argmin(axis=1))
    assert np.array_equal(counts.cpu().numpy(), np.bincount(sc.argmin(axis=1), minlength=10000))


def test_f16mx_split_k_mode_config4_against_the_oracle_and_across_buffer_sizes(ctx, capsys):
    """amx_ffnn_model.tuning ksplit=4 (opt-in, round 5): in a pass of at most 256 frames every 2048 x 2048 layer runs FOUR workgroups per
    tile, each over a quarter of K; the last one to arrive adds the four partial sums in group order -- a different association of the
    sum over k than the default's.  So
      (a) BASELINE config 4's network on 1024 frames, scored as the decoder would (fills of 256), against the f64-accumulating oracle on
          EVERY score -- the same assertions as the default mode;
      (b) within the mode a frame's scores do not depend on the fill it is scored in, as long as the fill is split at all: 256 frames at
          once = fills of 1, 100, 255 frames, bit for bit, run after run (the arrival order of the workgroups must not matter);
      (c) against the default order the scores differ by f32 rounding only, and a pass too large to be split (1024 frames) IS the default;
      (d) networks whose layers are too short to split and every activation still work"""
    import rasr_amd
    from oracle import oracle_ffnn_score
    dims = [440] + [2048] * 6 + [10000]
    Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
    T = 1024
    x = feats(T, 440, 6)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="f16mx", tuning="ksplit=4")
    got = np.concatenate([nn.score(x[t0:t0 + 256]) for t0 in range(0, T, 256)])
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=True)
    dflt = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="f16mx").score(x)
    rep = nn_parity_report(got, want, other=dflt, gap=1e-5)
    with capsys.disabled():
        print("\nf16mx ksplit=4 config 4 parity (fills of 256):", json.dumps(rep))
    assert rep["bar_violations"] == 0 and rep["worst_over_bar"] <= 0.5, rep
    assert rep["worst_pure_relative"] <= 1e-4, rep
    assert rep["argmin_mismatches_outside_gap_rule"] == 0, rep
    assert np.count_nonzero(got.view(np.uint32) != dflt.view(np.uint32)) > 0                 # another order of summation ...
    assert np.max(np.abs(got - dflt) / (np.abs(dflt) + 1.0)) < 2e-5                         # ... by rounding only
    assert np.array_equal(nn.score(x).view(np.uint32), dflt.view(np.uint32))                # 1024 frames at once: not split = the default
    for rep_ in range(3):
        for n, t0 in ((256, 0), (1, 5), (100, 37), (255, 1), (256, 768)):
            part = nn.score(x[t0:t0 + n])
            assert np.array_equal(part.view(np.uint32), got[t0:t0 + n].view(np.uint32)), (n, t0, rep_)
    # a fill that straddles two of the reference fills is still the same frames through the same split kernels
    assert np.array_equal(nn.score(x[100:300]).view(np.uint32), got[100:300].view(np.uint32))
    for d, act in (([440, 300, 33, 64, 50], 1), ([64, 96, 96, 10], 2), ([200, 2048, 2048, 40], 3)):
        W2, b2, a2, l2 = synth.ffnn(d, seed=11 + act, act=act)
        x2 = feats(200, d[0], 12)
        g2 = rasr_amd.NnBatchFeatureScorer(ctx, W2, b2, a2, log_prior=l2, precision="f16mx", tuning="ksplit=4").score(x2)
        w2 = oracle_ffnn_score(W2, b2, a2, x2, log_prior=l2, prior_scale=1.0, acc64=True)
        assert np.all(np.abs(g2 - w2) <= 1e-4 * np.abs(w2) + 1e-4), (d, np.abs(g2 - w2).max())
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.NnBatchFeatureScorer(ctx, Ws[:1], bs[:1], [0], precision="bf16", tuning="ksplit=4")
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.NnBatchFeatureScorer(ctx, Ws[:1], bs[:1], [0], precision="f16mx", tuning="ksplit=3")


@pytest.mark.parametrize("n_out", [2500, 2501])
def test_f16mx_tile_configurations_agree(ctx, monkeypatch, n_out):
    """every tile configuration (128x128, 128x64, 256x256) walks the K-tiles in the same order and issues f16 slab 0, f16 slab 1,
    scaled cross product per 32 x 32 block: scores, best states and accumulators are bit-identical; hidden layer through the
    block-writing epilogue of every configuration; 2501 makes the score rows unaligned (guarded stores)"""
    import torch

    import rasr_amd
    from oracle import oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([64, 300, n_out], seed=21)
    x = feats(8200, 64, 22)
    xd = torch.from_numpy(x).cuda()
    ctx.use_torch_stream()
    results = {}
    for cfg in ("0", "3", "6", "2", "4", "5", "7", "8", "9", "11", "12", "14"):   # 14: hidden layers on 64 x 64 tiles, four K-tiles per barrier (the default below half a tile of 128 x 64 per CU); 4 / 5 / 7 / 8 / 9: the 256 x 256 tile's K-loop variants (L2 prefetch, one issuing wave per SIMD, both, ping-pong halves, one self-pipelined wave per SIMD); 11: the one-tile-per-CU configuration without read-ahead (3 has it since round 6)
        nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="f16mx", tuning="tile=" + cfg)
        sc = torch.full((8200, n_out), float("nan"), dtype=torch.float32, device="cuda")
        best = torch.zeros(8200, dtype=torch.int32, device="cuda")
        counts = torch.zeros(n_out, dtype=torch.int64, device="cuda")
        ssum = torch.zeros(1, dtype=torch.float64, device="cuda")
        for _ in range(2):
            nn.score_stats_dev(xd, 64, 8200, sc, best, counts, ssum)
        torch.cuda.synchronize()
        results[cfg] = (sc.cpu().numpy(), best.cpu().numpy(), counts.cpu().numpy(), float(ssum.item()))
        plain = nn.score(x[:700])
        assert np.array_equal(plain.view(np.uint32), results[cfg][0][:700].view(np.uint32))
    ref = results["0"]
    assert np.isfinite(ref[0]).all()
    assert np.array_equal(ref[1], ref[0].argmin(axis=1))
    assert np.array_equal(ref[2], 2 * np.bincount(ref[1], minlength=n_out))
    for cfg in ("3", "6", "2", "4", "5", "7", "8", "9", "11", "12", "14"):
        got = results[cfg]
        assert np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32)), cfg
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), cfg
        assert abs(got[3] - ref[3]) <= 1e-9 * abs(ref[3]), cfg
    want = oracle_ffnn_score(Ws, bs, acts, x[:1500], log_prior=logp, prior_scale=1.0, acc64=True)
    assert np.all(np.abs(ref[0][:1500] - want) <= 1e-4 * np.abs(want) + 1e-4), np.abs(ref[0][:1500] - want).max()


def test_f16mx_full_size_shard_properties(ctx, capsys):
    """config 5 shard scale (40 000 frames: more than one internal pass, every layer on the 256 x 256 tiles): frame permutations
    permute the scores bit for bit, fused statistics equal a recount, a slice on the small-batch tiles equals the rows of the big
    pass, a row sample meets the bar of the f64 oracle (arg-min and pure relative error printed)"""
    import torch

    import rasr_amd
    from oracle import oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([440] + [2048] * 6 + [10000], seed=7)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="f16mx")
    T = 40000
    x = np.random.Generator(np.random.PCG64(300)).standard_normal((T, 440)).astype(np.float32)
    ctx.use_torch_stream()
    xd = torch.from_numpy(x).cuda()
    s = torch.empty((T, 10000), dtype=torch.float32, device="cuda")
    best = torch.empty((T,), dtype=torch.int32, device="cuda")
    counts = torch.zeros((10000,), dtype=torch.int64, device="cuda")
    ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
    nn.score_stats_dev(xd, 440, T, s, best, counts, ssum)
    torch.cuda.synchronize()
    perm = torch.randperm(T, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    s2 = torch.empty_like(s)
    nn.score_dev(xd[perm].contiguous(), 440, T, s2)
    torch.cuda.synchronize()
    assert torch.equal(s2.view(torch.int32), s[perm].view(torch.int32))
    del s2
    am = s.argmin(dim=1)
    assert torch.equal(best.long(), am)
    assert torch.equal(counts, torch.bincount(am, minlength=10000))
    ref_sum = float(s.gather(1, am[:, None]).double().sum())
    assert abs(float(ssum[0]) - ref_sum) <= 1e-9 * abs(ref_sum)
    s3 = torch.empty((1024, 10000), dtype=torch.float32, device="cuda")
    nn.score_dev(xd[32000:33024].contiguous(), 440, 1024, s3)
    torch.cuda.synchronize()
    assert torch.equal(s3.view(torch.int32), s[32000:33024].view(torch.int32))
    rows = np.r_[0:24, 32760:32776, 39990:40000]
    want = oracle_ffnn_score(Ws, bs, acts, x[rows], log_prior=logp, prior_scale=1.0, acc64=True)
    got = s[torch.from_numpy(rows).cuda()].cpu().numpy()
    rep = nn_parity_report(got, want, gap=1e-5)
    with capsys.disabled():
        print("\nf16mx shard sample parity:", json.dumps(rep))
    assert rep["bar_violations"] == 0 and rep["worst_pure_relative"] <= 1e-4 and rep["argmin_mismatches_outside_gap_rule"] == 0, rep


def test_f16mx_on_demand_hidden_activation(ctx):
    """amx_ffnn_forward_hidden_dev in this mode: the last hidden layer leaves as f32 through the score epilogue"""
    import torch

    import rasr_amd
    from oracle import oracle_ffnn_score
    for act in (1, 2, 3):
        Ws, bs, acts, logp = synth.ffnn([40, 96, 130, 50], seed=5, act=act)
        x = feats(300, 40, 9)
        nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="f16mx")
        ctx.use_torch_stream()
        xd = torch.from_numpy(x).cuda()
        hid = torch.empty((300, 130), dtype=torch.float32, device="cuda")
        nn.forward_hidden_dev(xd, 40, 300, hid)
        torch.cuda.synchronize()
        # hidden activation = the "scores" of the truncated network without prior, negated back, through the activation
        z = -oracle_ffnn_score(Ws[:2], bs[:2], [act, 0], x, acc64=True).astype(np.float64)
        want = {1: np.maximum(z, 0), 2: 1 / (1 + np.exp(-z)), 3: np.tanh(z)}[act]
        assert np.allclose(hid.cpu().numpy(), want, rtol=1e-4, atol=1e-4), act


def test_f16mx_values_outside_the_f16_range_fail_loudly(ctx):
    """a feature beyond 65504 cannot be represented: the call that met it reports it (host entry point) and the handle stays
    poisoned; weights beyond the range are refused at creation"""
    import rasr_amd
    Ws, bs, acts, logp = synth.ffnn([16, 32, 8], seed=3)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="f16mx")
    x = feats(10, 16, 1)
    assert np.isfinite(nn.score(x)).all()
    x[3, 5] = 1.0e6
    with pytest.raises(rasr_amd.AmxError):
        nn.score(x)
    with pytest.raises(rasr_amd.AmxError):
        nn.score(feats(10, 16, 2))
    Ws[0][1, 2] = 7.0e4
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="f16mx")


@pytest.mark.parametrize("bad", ["feature", "inf", "nan", "hidden"])
def test_f16mx_device_entry_points_fail_in_the_pass_that_overflows(ctx, bad):
    """the *_dev entry points only enqueue work; amx_ffnn_wait_dev behind a pass reports THAT pass (the pass that
    overflowed used to return AMX_OK and the NEXT call failed).  A feature beyond the range, inf, NaN (fmaxf would drop it), and a hidden
    activation beyond the range (weights and inputs in range, their product not); afterwards every entry point that would consume the
    handle's state refuses: score_dev, forward_hidden_dev, on-demand scoring"""
    import torch

    import rasr_amd
    Ws, bs, acts, logp = synth.ffnn([16, 32, 8], seed=3)
    x = feats(300, 16, 1)
    if bad == "hidden":
        Ws[0][:] = 900.0          # 16 x 900 x |x| ~ 1e4 .. 1e5 after the ReLU: beyond 65504 for some frames
        x = np.abs(x) * 8
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="f16mx")
    ctx.use_torch_stream()
    ok = torch.from_numpy(feats(300, 16, 9) * 1e-3 if bad == "hidden" else feats(300, 16, 9)).cuda().float()
    sc = torch.empty((300, 8), dtype=torch.float32, device="cuda")
    nn.score_dev(ok, 16, 300, sc)
    nn.wait_dev()                                       # a clean pass: OK
    assert torch.isfinite(sc).all()
    if bad == "feature":
        x[7, 3] = 1.0e6
    elif bad == "inf":
        x[7, 3] = np.inf
    elif bad == "nan":
        x[7, 3] = np.nan
    nn.score_dev(torch.from_numpy(x).cuda(), 16, 300, sc)    # enqueues: AMX_OK
    with pytest.raises(rasr_amd.AmxError) as e:
        nn.wait_dev()                                   # ... and THIS pass fails
    assert e.value.status == -4 and "f16 range" in str(e.value)
    with pytest.raises(rasr_amd.AmxError):
        nn.score_dev(ok, 16, 300, sc)
    hid = torch.empty((300, 32), dtype=torch.float32, device="cuda")
    with pytest.raises(rasr_amd.AmxError):
        nn.forward_hidden_dev(ok, 16, 300, hid)
    # the same inputs through split bf16: no limit
    got = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="bf16x3").score(np.nan_to_num(x, nan=0.0, posinf=1e6))
    assert np.isfinite(got).all()


@pytest.mark.parametrize("family", __import__("tests.ffnn_families", fromlist=["FAMILIES"]).FAMILIES)
def test_f16mx_operand_families_that_are_not_gaussian(ctx, family, capsys):
    """per-block outliers (one value 2^8 .. 2^12 times the other 31 of its exponent block) in features, weights or both, log-normal
    weight rows, unnormalised MFCC context windows (c0 beside c30), all-positive operands (nothing cancels), scaled operands, sparse
    post-ReLU activations: every score within the 1e-4 |ref| + 1e-4 bar of the f64-accumulating oracle, best state identical on every
    frame the 1e-5 gap rule keeps; prints worst-over-bar per family (profiles/r05/f16mx_families.log)"""
    import rasr_amd
    from oracle import oracle_ffnn_score
    from tests.ffnn_families import make
    from tests.parity import nn_parity_report
    dims = [440, 768, 768, 1500]
    Ws, bs, acts, logp, x = make(family, dims, 384, 300 + len(family))
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=True)
    assert np.isfinite(want).all()
    raw = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="f16mx", tuning="mx_fallback=off")   # the scheme itself
    assert raw.effective_precision()[0] == "f16mx"
    got = raw.score(x)
    f32 = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="fp32").score(x)
    rep = nn_parity_report(got, want, other=f32, gap=1e-5)
    sx3 = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="bf16x3").score(x)
    x3 = nn_parity_report(sx3, want, gap=1e-5)
    # what a caller gets by default: heavy-tailed weights (block-maximum statistic above 4) compute in split bf16
    dflt = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="f16mx")
    eff, ratio = dflt.effective_precision()
    heavy = family in ("block-outliers-w", "block-outliers-both", "lognormal-rows")
    assert eff == ("bf16x3" if heavy else "f16mx"), (family, eff, ratio)
    assert (ratio > 4.0) == heavy, (family, ratio)
    if heavy:
        assert np.array_equal(dflt.score(x).view(np.uint32), sx3.view(np.uint32))
        with pytest.raises(rasr_amd.AmxError, match="ksplit"):   # an f16mx schedule on a handle that would compute in split bf16: refused, not ignored
            rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="f16mx", tuning="ksplit=4")
    r32 = nn_parity_report(f32, want, gap=1e-5)
    # Ill-conditioned families (outliers of 2^8 .. 2^12, log-normal rows, features of hundreds): a score is then a difference of terms
    # 10^3 .. 10^5 times larger than itself, and f32 ACCUMULATION -- the reference's own sgemm -- is off by 2-5 times the bar there
    # (the exact-f32 MFMA path, measured beside it).  Measured on every family: the f16mx error is a constant ~27 x the error of f32
    # accumulation on the same network (split bf16 ~10 x), outliers or not -- the exponent blocks do not break on outliers, the bar breaks
    # on conditioning.  So: the strict bar wherever f32 arithmetic itself meets it; elsewhere the bar relative to the frame's score scale
    # (1e-4 of the largest |score| of the frame: what a decoder compares are scores of one frame) and within 40 x the f32 path's error.
    strict = r32["worst_over_bar"] <= 1.0
    err = np.abs(got.astype(np.float64) - want)
    rowscale = np.abs(want).max(axis=1, keepdims=True)
    norm_bar = 1e-4 * np.maximum(np.abs(want), rowscale) + 1e-4
    worst_norm = float((err / norm_bar).max())
    with capsys.disabled():
        print("\n[f16mx family %-20s] block ratio %.2f -> default runs %s | worst/bar: f16mx %.4f  bf16x3 %.4f  exact-f32 MFMA %.4f  (%s) | f16mx / frame-scale bar %.4f | f16mx pure rel %.2e, "
              "arg-min mismatches %d (outside the gap rule %d, vs exact f32 %d) | scores %.3g .. %.3g"
              % (family, ratio, eff, rep["worst_over_bar"], x3["worst_over_bar"], r32["worst_over_bar"], "strict bar applies" if strict else "f32 itself misses the strict bar",
                 worst_norm, rep["worst_pure_relative"], rep["argmin_mismatches"], rep["argmin_mismatches_outside_gap_rule"],
                 rep["argmin_mismatches_vs_fp32_mfma"], float(want.min()), float(want.max())))
    if strict:
        assert rep["bar_violations"] == 0, rep
    else:
        # the scheme's error is BOUNDED on outliers (one value carrying every block: ~115 x f32's), and the default handle does not
        # run it on such weights at all
        assert worst_norm <= 2.0 and rep["worst_over_bar"] <= 150.0 * r32["worst_over_bar"], (worst_norm, rep, r32)
    assert rep["argmin_mismatches_outside_gap_rule"] == 0, rep
