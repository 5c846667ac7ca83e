"""GPU: committed golden vectors, the context window, the epoch accumulators and the whole chain."""
import os

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_vectors(ctx):
    import rasr_amd
    z = np.load(os.path.join(GOLD, "orc_mfcc.npz"))
    pcm = z["pcm_s16"].astype(np.float32)
    for tag, kw in (("mfcc16", dict(nr_cepstrum_coefficients=16)), ("mfcc40", dict(nr_cepstrum_coefficients=40, filter_width=138.0))):
        got = rasr_amd.MfccExtractor(ctx, **kw).run(pcm)
        assert np.all(np.abs(got - z[tag]) <= 1e-4 * np.abs(z[tag]) + 1e-4)
    g = np.load(os.path.join(GOLD, "orc_gmm.npz"))
    model = {k[6:]: g[k] for k in g.files if k.startswith("model_")}
    model["dim"] = int(model["dim"])
    sc, best = rasr_amd.GmmFeatureScorer(ctx, model).score(g["feats"])
    assert np.array_equal(sc.view(np.uint32), g["max_scores"].view(np.uint32)) and np.array_equal(best, g["max_best"])
    ss, _ = rasr_amd.GmmFeatureScorer(ctx, model, feature_scorer_type="diagonal-sum").score(g["feats"])
    assert np.allclose(ss, g["sum_scores"], rtol=1e-5, atol=1e-5)
    f = np.load(os.path.join(GOLD, "orc_ffnn.npz"))
    Ws, bs = [f["W%d" % i] for i in range(3)], [f["b%d" % i] for i in range(3)]
    got = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, [1, 1, 0], log_prior=f["log_prior"], precision="fp32").score(f["feats"])
    assert np.array_equal(got.view(np.uint32), f["scores_fma"].view(np.uint32))
    assert np.all(np.abs(got - f["scores64"]) <= 1e-4 * np.abs(f["scores64"]) + 1e-4)


def test_context_window_copy_margin(ctx):
    import torch

    import rasr_amd
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=8)
    lens = np.array([400, 2000, 161 * 7, 5000])          # 1, 11, 6, 30 frames
    off = np.concatenate([[0], np.cumsum(lens)])
    plan = fe.plan(off)
    F = plan.total_frames
    x = np.random.Generator(np.random.PCG64(1)).standard_normal((F, 8)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    out = torch.empty((F, 72), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    ctx.context_window(plan, xd, 8, 3, 2, out, 72)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for u in range(len(lens)):
        s0, s1 = plan.frame_offsets[u], plan.frame_offsets[u + 1]
        for t in range(s0, s1):
            idx = np.clip(np.arange(t - 3, t + 3), s0, s1 - 1)
            assert np.array_equal(got[t, :48], x[idx].reshape(-1))
            assert np.all(got[t, 48:] == 0)


def test_epoch_accumulators(ctx):
    import torch
    rng = np.random.Generator(np.random.PCG64(5))
    T, M = 1000, 777
    sc = rng.standard_normal((T, M)).astype(np.float32)
    sc[10, 5] = sc[10, 700] = -50.0                     # tie: the smaller state index wins
    sc[11] = 3.0                                        # all equal: state 0
    d = torch.from_numpy(sc).cuda()
    best = torch.empty(T, dtype=torch.int32, device="cuda")
    counts = torch.zeros(M, dtype=torch.int64, device="cuda")
    ssum = torch.zeros(1, dtype=torch.float64, device="cuda")
    ctx.use_torch_stream()
    ctx.stats_accumulate(d, T, M, best, counts, ssum)
    ctx.stats_accumulate(d, T, M, best, counts, ssum)   # accumulates
    torch.cuda.synchronize()
    want = sc.argmin(axis=1)
    assert np.array_equal(best.cpu().numpy(), want) and want[10] == 5 and want[11] == 0
    assert np.array_equal(counts.cpu().numpy(), 2 * np.bincount(want, minlength=M))
    assert abs(float(ssum[0]) - 2 * sc.min(axis=1).astype(np.float64).sum()) < 1e-6


def test_chain_audio_to_best_state(ctx):
    """audio -> MFCC -> GMM scores -> best state, against the oracle chain; the state decision is identical
    wherever the oracle's top-2 margin exceeds the front-end tolerance."""
    import rasr_amd
    from oracle import OracleGmm, OracleMfcc
    pcm = synth.waveform(40000, seed=77)
    model = synth.gmm_cart(500, 1, 4, 16, seed=78, pooled=False)
    ceps = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=16).run(pcm)
    sc, _ = rasr_amd.GmmFeatureScorer(ctx, model).score(ceps)
    oc = OracleMfcc(n_ceps=16).run(pcm)
    osc, _ = OracleGmm(model).score(oc)
    part = np.partition(osc, 1, axis=1)
    clear = (part[:, 1] - part[:, 0]) > 1e-2
    assert clear.mean() > 0.9
    assert np.array_equal(sc.argmin(1)[clear], osc.argmin(1)[clear])
    assert np.allclose(sc, osc, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("T", [1, 300, 1025])
def test_fused_best_state_statistics(ctx, precision, T):
    """arg-min fused into the output-layer epilogue == arg-min over the score matrix it wrote (first minimum wins)"""
    import torch

    import rasr_amd
    Ws, bs, acts, logp = synth.ffnn([40, 300, 1000], seed=5)
    Ws[-1][7] = Ws[-1][900]                 # states 7 and 900 tie on every frame
    bs[-1][7] = bs[-1][900]
    logp[7] = logp[900]
    nn = rasr_amd.NnBatchFeatureScorer(ctx, Ws, bs, acts, log_prior=logp, precision=precision)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(2)).standard_normal((T, 40)).astype(np.float32)).cuda()
    sc = torch.empty((T, 1000), dtype=torch.float32, device="cuda")
    best = torch.full((T,), -1, dtype=torch.int32, device="cuda")
    counts = torch.zeros(1000, dtype=torch.int64, device="cuda")
    ssum = torch.zeros(1, dtype=torch.float64, device="cuda")
    ctx.use_torch_stream()
    nn.score_stats_dev(x, 40, T, sc, best, counts, ssum)
    torch.cuda.synchronize()
    s = sc.cpu().numpy()
    want = s.argmin(axis=1)
    assert np.array_equal(best.cpu().numpy(), want)
    assert 900 not in want
    assert np.array_equal(counts.cpu().numpy(), np.bincount(want, minlength=1000))
    assert abs(float(ssum[0]) - s.min(axis=1).astype(np.float64).sum()) < 1e-6 * max(1.0, abs(float(ssum[0])))


def test_feature_cache_between_front_end_and_scorer(ctx, tmp_path):
    """feature-extraction job writes a cache, the scoring job reads it: scores are bit-identical to
    scoring the extractor's output directly, and the cache entry is the reference's block layout."""
    import rasr_amd
    from oracle import cache_format as cf
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=16)
    model = synth.gmm_cart(50, 2, 6, 16, seed=31, pooled=False)
    gmm = rasr_amd.GmmFeatureScorer(ctx, model)
    path = str(tmp_path / "mfcc.cache")
    direct = {}
    with rasr_amd.FileArchive(path, "w") as a:
        for s, n in enumerate((16000, 4000, 23456)):
            pcm = synth.waveform(n, seed=40 + s)
            x = fe.run(pcm)
            start = np.array([fe.frame_start_time(i) for i in range(len(x))])
            end = np.minimum(start + 0.025, n / 16000.0)
            a.write_features("corpus/rec/%d" % s, x, np.stack([start, end], 1), compress=bool(s % 2),
                             attributes={"sample-rate": "100", "datatype": "vector-f32"})
            direct["corpus/rec/%d" % s] = (x, gmm.score(x))
    with rasr_amd.FileArchive(path) as a:
        for seg, (x, (sc, best)) in direct.items():
            rx, rt = a.read_features(seg)
            assert rx.tobytes() == x.tobytes()
            assert a.read_file(seg) == cf.entry_payload(x, rt)
            assert a.read_attributes(seg)["datatype"] == "vector-f32"
            sc2, best2 = gmm.score(rx)
            assert np.array_equal(sc2.view(np.uint32), sc.view(np.uint32)) and np.array_equal(best2, best)


def test_streamed_ingest_delivers_the_partitions_utterances_bit_for_bit(ctx):
    """bench.py --ingest streamed (StreamedIngest: pinned s16 corpus window -> hipMemcpyAsync on a copy stream -> two HBM slots ->
    amx_mfcc_run_plan_dev_s16 behind an event): over more steps than there are slots and more than the host window holds, every
    step's cepstra equal the cepstra of exactly the utterances the rank's CorpusWalker names -- computed from the samples directly,
    as f32, in one call -- bit for bit; two ranks' lists are disjoint and lie in their partitions."""
    import argparse

    import torch

    import bench
    import rasr_amd
    ctx.use_torch_stream()
    args = argparse.Namespace(utt_seconds=1.0, utterances=6, corpus_hours=0.05, ingest="streamed")   # 180 utterances of 1 s
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=13)
    n = 16000
    base = synth.waveform(n + 8192, seed=9)
    seen = {}
    for rank in (0, 1):
        ing = bench.StreamedIngest(args, rank, 2, n_steps=3)           # host window: 5 batches; 9 steps wrap it
        plan = fe.plan(np.arange(args.utterances + 1, dtype=np.int64) * n)
        ceps = torch.empty((plan.total_frames, 13), dtype=torch.float32, device="cuda")
        for step in range(9):
            pcm = ing.take()
            assert pcm.dtype == torch.int16
            fe.run_plan(plan, pcm, ceps)
            ing.release()
            ids = ing.visited[-1]
            assert len(ids) == args.utterances and all(int(g) % 2 == rank for g in ids)
            if step < ing.nb:   # inside the host window the samples are the named utterances'; beyond it the window wraps (bench note)
                want_pcm = np.concatenate([base[int(g) % 8192:int(g) % 8192 + n] for g in ids])
                want = torch.empty_like(ceps)
                fe.run_plan(plan, torch.from_numpy(want_pcm).cuda(), want)
                torch.cuda.synchronize()
                assert torch.equal(ceps.view(torch.int32), want.view(torch.int32)), (rank, step)
        torch.cuda.synchronize()
        seen[rank] = np.concatenate(ing.visited)
        assert len(np.unique(seen[rank])) == len(seen[rank]) == 9 * args.utterances
    assert not set(seen[0].tolist()) & set(seen[1].tolist())


@pytest.mark.probe
def test_device_clock_samples(ctx):
    """measurement entry points (bench.py's shader clock; a PROBE, not a parity test: marked `probe`, collected last by
    tests/conftest.py, and it asserts only what the hardware promises).  s_memrealtime is ONE 100 MHz counter for the chip: two samples
    in stream order, milliseconds apart, are ordered.  s_memtime is a counter PER CU and the CUs' counters are not aligned: two samples
    may land on different CUs (also of one XCD, also for the single-workgroup form -- round 5's driver box returned a "negative"
    interval of 3.9e7 ticks), so it is held to `> 0` and nothing else.  The per-XCD form files every sample under the XCC id it read
    (amx_device_clocks_dev / amx_device_clocks_xcd_dev)."""
    import torch
    ctx.use_torch_stream()
    one = torch.zeros((2, 2), dtype=torch.int64, device="cuda")
    xcd = torch.zeros((2, 8, 2), dtype=torch.int64, device="cuda")
    ctx.device_clocks(one[0])
    ctx.device_clocks_xcd(xcd[0])
    a = torch.randn((2048, 2048), device="cuda")
    for _ in range(20):
        a = (a @ a).clamp_(-1, 1)   # a few milliseconds of work between the samples
    ctx.device_clocks(one[1])
    ctx.device_clocks_xcd(xcd[1])
    torch.cuda.synchronize()
    o, x = one.cpu().numpy(), xcd.cpu().numpy()
    assert o[1, 1] > o[0, 1] > 0          # s_memrealtime: ordered
    assert (o[:, 0] > 0).all()            # s_memtime: sampled, no order promised
    reached = (x[0, :, 1] > 0) & (x[1, :, 1] > 0)
    assert reached.sum() >= 1, x          # placement is not promised (eight of eight observed)
    assert (x[1, :, 1][reached] > x[0, :, 1][reached]).all()   # the 100 MHz counter is one clock for the chip: ordered
    assert (x[:, :, 0][:, reached] > 0).all()
