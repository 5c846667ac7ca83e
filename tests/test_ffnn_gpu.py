"""GPU parity: feed-forward NN scorer (MFMA GEMM chain) through the C ABI against the oracle."""
import json
import os

import numpy as np
import pytest

from tests import synth
from tests.parity import nn_parity_report

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def feats(T, dim, seed):
    return np.random.Generator(np.random.PCG64(seed)).standard_normal((T, dim)).astype(np.float32)


def softmax(z):
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def test_reference_unit_test_vectors(ctx):
    """Known answers of the reference's own tests (Test/Nn_LinearAndActivationLayer.cc:78-104,157-189;
    Test/Nn_NeuralNetwork.cc:38-73,104-119), transcribed into tests/golden/nn_kat.json."""
    import rasr_amd
    kat = json.load(open(os.path.join(GOLD, "nn_kat.json")))
    for case in kat["cases"]:
        Ws = [np.array(w, np.float32) for w in case["W"]]
        bs = [np.array(b, np.float32) for b in case["bias"]]
        acts = case["hidden_activation"] + [0]
        x = np.array(case["input"], np.float32)
        nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, precision="fp32")
        z = -nn.score(x)                              # scorer returns -(Wx+b)
        if "linear" in case:
            assert np.allclose(z, np.array(case["linear"]), atol=1e-5)
        if "softmax" in case:
            assert np.allclose(softmax(z.astype(np.float64)), np.array(case["softmax"]), atol=case["tol"])
        if "sigmoid" in case:
            assert np.allclose(1 / (1 + np.exp(-z.astype(np.float64))), np.array(case["sigmoid"]), atol=case["tol"])


@pytest.mark.parametrize("T", [1, 100, 128, 129, 1024])
def test_fp32_path_matches_cpu(ctx, T):
    """fp32 MFMA path: <= 1e-4 relative on the scores (BASELINE north star), argmin state identical."""
    import rasr_amd
    from oracle import oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([440, 256, 256, 1000], seed=7)
    x = feats(T, 440, 6)
    got = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="fp32").score(x)
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=True)
    assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-4), np.abs(got - want).max()
    assert np.array_equal(got.argmin(axis=1), want.argmin(axis=1))


def test_fp32_path_is_a_k_ordered_fma_chain(ctx):
    """v_mfma_f32_32x32x2_f32 accumulates like fmaf(a_k, b_k, acc) in ascending k: scores are bit-identical
    to the oracle's fmaf-chain mode (including hidden layers, K padding and ragged tiles)."""
    import rasr_amd
    from oracle import oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([100, 70, 130, 37], seed=3)
    x = feats(133, 100, 4)
    got = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="fp32").score(x)
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, acc64=2)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()


@pytest.mark.parametrize("act", [1, 2, 3])
def test_activations(ctx, act):
    import rasr_amd
    from oracle import oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([64, 130, 77], seed=17, act=act)
    x = feats(50, 64, 18)
    got = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=0.6, precision="fp32").score(x)
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=0.6, acc64=True)
    assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-4), np.abs(got - want).max()


def test_bf16_path_accuracy(ctx):
    """bf16 inputs / f32 accumulate: not a 1e-4 path.  Bound: error relative to the score scale < 2e-2 and the
    result equals an emulation that rounds weights and layer inputs to bf16 (what the kernel computes)."""
    import rasr_amd
    import torch
    Ws, bs, acts, logp = synth.ffnn([440, 512, 512, 1000], seed=7)
    x = feats(300, 440, 6)
    got = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="bf16").score(x)
    # emulation in f64 with bf16-rounded operands
    bf = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float64).numpy()
    a = bf(x)
    for l, (W, b) in enumerate(zip(Ws, bs)):
        z = a @ bf(W).T
        if l == len(Ws) - 1:
            z = z + (b - np.float32(1.0) * logp).astype(np.float64)
            emu = -z
        else:
            a = bf(np.maximum(z + b, 0).astype(np.float32))
    assert np.allclose(got, emu, rtol=2e-3, atol=2e-3), np.abs(got - emu).max()
    from oracle import oracle_ffnn_score
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, acc64=True)
    scale = np.abs(want).mean()
    assert np.abs(got - want).max() < 5e-2 * scale


def test_config4_shape_bf16_properties(ctx):
    """BASELINE config 4 (440 -> 6x2048 -> 10000, batch 1024): linearity-free properties -- rows are
    independent (scoring a subset gives identical rows) and the prior shifts scores exactly."""
    import rasr_amd
    Ws, bs, acts, logp = synth.ffnn([440] + [2048] * 6 + [10000], seed=7)
    x = feats(1024, 440, 6)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="bf16")
    full = nn.score(x)
    assert full.shape == (1024, 10000) and np.isfinite(full).all()
    part = nn.score(x[100:300])
    assert np.array_equal(part.view(np.uint32), full[100:300].view(np.uint32))
    nn0 = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=None, precision="bf16")
    base = nn0.score(x[:64])
    # score = -(z + b - logp) = base + logp; the bias is added in f32 in the epilogue
    assert np.allclose(full[:64] - base, logp[None, :], atol=1e-3)


@pytest.mark.parametrize("n_out", [2500, 2501])
def test_tile_configurations_agree(ctx, monkeypatch, n_out):
    """Every tile configuration of the bf16 GEMM (128x128, 128x64, 256x256, and the cross-tile pipelined 256x256 kernel
    the large batches use) accumulates in the same k order, so scores, best states and accumulators are bit-identical.
    Shape chosen so that workgroups of the persistent kernels walk several tiles, including frame- and state-edge tiles
    (8200 = 32*256 + 8 frames; 2500 = 9*256 + 196 states; 2501 makes the score rows unaligned -> guarded stores)."""
    import torch

    import rasr_amd
    Ws, bs, acts, logp = synth.ffnn([64, 300, n_out], seed=21)
    x = feats(8200, 64, 22)
    xd = torch.from_numpy(x).cuda()
    ctx.use_torch_stream()
    results = {}
    for cfg in ("0", "3", "6", "4", "2", "11"):   # 11: tile 6 with loader waves for the hidden layers (round 6, opt-in)
        nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="bf16", tuning="tile=" + cfg)
        sc = torch.full((8200, n_out), float("nan"), dtype=torch.float32, device="cuda")
        best = torch.zeros(8200, dtype=torch.int32, device="cuda")
        counts = torch.zeros(n_out, dtype=torch.int64, device="cuda")
        ssum = torch.zeros(1, dtype=torch.float64, device="cuda")
        for _ in range(2):  # second pass: the counted-wait path of the pipelined kernel in steady state
            nn.score_stats_dev(xd, 64, 8200, sc, best, counts, ssum)
        torch.cuda.synchronize()
        results[cfg] = (sc.cpu().numpy(), best.cpu().numpy(), counts.cpu().numpy(), float(ssum.item()))
        plain = nn.score(x[:700])
        assert np.array_equal(plain.view(np.uint32), results[cfg][0][:700].view(np.uint32))
    ref = results["0"]
    assert np.isfinite(ref[0]).all()
    assert np.array_equal(ref[1], ref[0].argmin(axis=1))
    assert np.array_equal(ref[2], 2 * np.bincount(ref[1], minlength=n_out))
    for cfg in ("3", "6", "4", "2", "11"):
        got = results[cfg]
        assert np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32)), cfg
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), cfg
        assert abs(got[3] - ref[3]) <= 1e-9 * abs(ref[3]), cfg


@pytest.mark.parametrize("graph", ["graph=1", None])
def test_small_batch_graph_replay(ctx, graph):
    """tuning graph=1: batches of <= 4096 frames on unchanged device buffers are replayed as a HIP graph from the third call on: the
    results must follow the CURRENT buffer contents, and the fused statistics must keep accumulating (None: the default since round 6,
    plain launches)"""
    import torch

    import rasr_amd
    Ws, bs, acts, logp = synth.ffnn([64, 256, 256, 1000], seed=31)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="bf16", tuning=graph)
    T = 300
    xd = torch.empty((T, 64), dtype=torch.float32, device="cuda")
    sc = torch.empty((T, 1000), dtype=torch.float32, device="cuda")
    best = torch.empty((T,), dtype=torch.int32, device="cuda")
    counts = torch.zeros((1000,), dtype=torch.int64, device="cuda")
    ssum = torch.zeros((1,), dtype=torch.float64, device="cuda")
    ctx.use_torch_stream()
    total = np.zeros(1000, np.int64)
    for rep in range(5):
        x = feats(T, 64, 40 + rep)
        xd.copy_(torch.from_numpy(x))
        nn.score_stats_dev(xd, 64, T, sc, best, counts, ssum)
        torch.cuda.synchronize()
        want = nn.score(x)                                  # host path: different buffers, plain launches
        got = sc.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), rep
        assert np.array_equal(best.cpu().numpy(), want.argmin(axis=1)), rep
        total += np.bincount(want.argmin(axis=1), minlength=1000)
        assert np.array_equal(counts.cpu().numpy(), total), rep
    for rep in range(4):                                    # plain scoring on the same buffers: its own graph
        x = feats(T, 64, 50 + rep)
        xd.copy_(torch.from_numpy(x))
        nn.score_dev(xd, 64, T, sc)
        torch.cuda.synchronize()
        assert np.array_equal(sc.cpu().numpy().view(np.uint32), nn.score(x).view(np.uint32)), rep


def test_full_size_shard_properties(ctx):
    """BASELINE config 5 shard scale (440 -> 6 x 2048 -> 10 000 in bf16, 40 000 frames: more than one internal pass of 32 768):
    rows are independent -- scoring the frames in another order permutes the scores bit for bit, and the fused arg-min
    statistics equal a recount from the score matrix; a small slice scored on its own equals the rows of the big pass."""
    import torch

    import rasr_amd
    Ws, bs, acts, logp = synth.ffnn([440] + [2048] * 6 + [10000], seed=7)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="bf16")
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
    am = s.argmin(dim=1)
    assert torch.equal(best.long(), am)
    assert torch.equal(counts, torch.bincount(am, minlength=10000))
    ref_sum = float(s.gather(1, am[:, None]).double().sum())
    assert abs(float(ssum[0]) - ref_sum) <= 1e-9 * abs(ref_sum)
    s3 = torch.empty((1024, 10000), dtype=torch.float32, device="cuda")
    nn.score_dev(xd[32000:33024].contiguous(), 440, 1024, s3)     # small-batch tile configuration, straddling the pass boundary
    torch.cuda.synchronize()
    assert torch.equal(s3.view(torch.int32), s[32000:33024].view(torch.int32))


# ---- split-bf16 ("bf16x3") precision: three bf16 MFMA products per f32 product, f32 accumulation

@pytest.mark.parametrize("T", [1, 100, 129, 1024])
def test_bf16x3_path_meets_the_fp32_bar(ctx, T):
    """the same <= 1e-4 relative (+1e-4 absolute) bar as the fp32 path, arg-min state identical, ragged shapes"""
    import rasr_amd
    from oracle import oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([440, 256, 300, 1000], seed=7)
    x = feats(T, 440, 6)
    got = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="bf16x3").score(x)
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=True)
    assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-4), np.abs(got - want).max()
    assert np.array_equal(got.argmin(axis=1), want.argmin(axis=1))


@pytest.mark.parametrize("act", [1, 2, 3])
def test_bf16x3_activations(ctx, act):
    import rasr_amd
    from oracle import oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([64, 130, 77], seed=17, act=act)
    x = feats(50, 64, 18)
    got = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=0.6, precision="bf16x3").score(x)
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=0.6, acc64=True)
    assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-4), np.abs(got - want).max()


def test_bf16x3_config4_full_size_against_the_oracle(ctx, capsys):
    """BASELINE config 4 at full size -- 440-6x2048-10000, batch 1024 -- against the f64-accumulating oracle on EVERY score (not a
    property test): |delta| <= 1e-4 |ref| + 1e-4 AND the pure relative error over |ref| > 1e-2 below 1e-4; the arg-min state equals
    the oracle's and the exact-f32 MFMA path's on ALL 1024 frames (no gap rule needed: the report prints how many frames a 1e-5
    rule would exclude); the fused statistics agree with a recount of the scores"""
    import torch

    import rasr_amd
    from oracle import oracle_ffnn_score
    dims = [440] + [2048] * 6 + [10000]
    Ws, bs, acts, logp = synth.ffnn(dims, seed=7)
    T = 1024
    x = feats(T, 440, 6)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="bf16x3")
    got = nn.score(x)
    want = oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, prior_scale=1.0, acc64=True)
    f32 = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="fp32").score(x)
    rep = nn_parity_report(got, want, other=f32, gap=1e-5)
    with capsys.disabled():
        print("\nbf16x3 config 4 parity:", json.dumps(rep))
    assert rep["bar_violations"] == 0 and rep["worst_over_bar"] <= 0.5 and rep["worst_pure_relative"] <= 1e-4, rep
    assert rep["argmin_mismatches_outside_gap_rule"] == 0 and rep["frames_excluded_by_gap_rule"] <= 0.005 * T, rep
    assert rep["argmin_mismatches"] <= rep["frames_excluded_by_gap_rule"] and rep["argmin_mismatches_vs_fp32_mfma"] <= rep["frames_excluded_by_gap_rule"], rep
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


@pytest.mark.parametrize("n_out", [2500, 2501])
def test_bf16x3_tile_configurations_agree(ctx, monkeypatch, n_out):
    """split bf16: every tile configuration -- 128x128, 128x64 (two and three stages), 256x256 and the cross-tile pipelined 256x256
    kernel with its two-barrier K-loop -- stages the same four planes and issues hi.hi, lo.hi, hi.lo in the same order, so scores,
    best states and accumulators are bit-identical; and they meet the 1e-4 bar of the oracle.  Shape as in the bf16 test: persistent
    workgroups walk several tiles incl. frame- and state-edge tiles, 2501 makes the score rows unaligned (guarded stores); the
    hidden layer (64 -> 300 -> n_out) goes through the fused activation + split epilogue of every configuration."""
    import torch

    import rasr_amd
    from oracle import oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([64, 300, n_out], seed=21)
    x = feats(8200, 64, 22)
    xd = torch.from_numpy(x).cuda()
    ctx.use_torch_stream()
    results = {}
    for cfg in ("0", "3", "6", "4", "2"):
        nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision="bf16x3", tuning="tile=" + cfg)
        sc = torch.full((8200, n_out), float("nan"), dtype=torch.float32, device="cuda")
        best = torch.zeros(8200, dtype=torch.int32, device="cuda")
        counts = torch.zeros(n_out, dtype=torch.int64, device="cuda")
        ssum = torch.zeros(1, dtype=torch.float64, device="cuda")
        for _ in range(2):  # second pass: the counted-wait path of the pipelined kernel in steady state
            nn.score_stats_dev(xd, 64, 8200, sc, best, counts, ssum)
        torch.cuda.synchronize()
        results[cfg] = (sc.cpu().numpy(), best.cpu().numpy(), counts.cpu().numpy(), float(ssum.item()))
        plain = nn.score(x[:700])
        assert np.array_equal(plain.view(np.uint32), results[cfg][0][:700].view(np.uint32))
    ref = results["0"]
    assert np.isfinite(ref[0]).all()
    assert np.array_equal(ref[1], ref[0].argmin(axis=1))
    assert np.array_equal(ref[2], 2 * np.bincount(ref[1], minlength=n_out))
    for cfg in ("3", "6", "4", "2"):
        got = results[cfg]
        assert np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32)), cfg
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), cfg
        assert abs(got[3] - ref[3]) <= 1e-9 * abs(ref[3]), cfg
    want = oracle_ffnn_score(Ws, bs, acts, x[:1500], log_prior=logp, prior_scale=1.0, acc64=True)
    assert np.all(np.abs(ref[0][:1500] - want) <= 1e-4 * np.abs(want) + 1e-4), np.abs(ref[0][:1500] - want).max()


def test_bf16x3_full_size_shard_properties(ctx, capsys):
    """BASELINE config 5 shard scale in split bf16 (440 -> 6 x 2048 -> 10 000, 40 000 frames: more than one internal pass of 32 768,
    every layer on the pipelined kernel): rows are independent -- scoring the frames in another order permutes the scores bit for
    bit --, the fused arg-min statistics equal a recount, a slice scored with the small-batch tiles equals the rows of the big pass,
    and a sample of rows meets the 1e-4 bar of the f64-accumulating oracle."""
    import torch

    import rasr_amd
    from oracle import oracle_ffnn_score
    Ws, bs, acts, logp = synth.ffnn([440] + [2048] * 6 + [10000], seed=7)
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, priori_scale=1.0, precision="bf16x3")
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
    nn.score_dev(xd[32000:33024].contiguous(), 440, 1024, s3)     # small-batch tile configuration, straddling the pass boundary
    torch.cuda.synchronize()
    assert torch.equal(s3.view(torch.int32), s[32000:33024].view(torch.int32))
    rows = np.r_[0:24, 32760:32776, 39990:40000]                  # first tile, the pass boundary, the ragged last tile
    want = oracle_ffnn_score(Ws, bs, acts, x[rows], log_prior=logp, prior_scale=1.0, acc64=True)
    got = s[torch.from_numpy(rows).cuda()].cpu().numpy()
    rep = nn_parity_report(got, want, gap=1e-5)
    with capsys.disabled():
        print("\nbf16x3 shard sample parity:", json.dumps(rep))
    assert rep["bar_violations"] == 0 and rep["worst_pure_relative"] <= 1e-4 and rep["argmin_mismatches_outside_gap_rule"] == 0, rep
