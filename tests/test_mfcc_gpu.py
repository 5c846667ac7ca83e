"""GPU parity: fused MFCC kernel (through the C ABI) against the oracle restatement of mfcc.flow."""
import os

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

# Tolerance: cepstra are sums of n_filters log10 energies (|c0| ~ 100, higher ones O(1)).  The device FFT
# uses f32 fmaf butterflies with table twiddles, the reference f64-recurrence twiddles with f32 data, so
# spectra agree to ~1e-6 relative; after log10 and the DCT that is <= 1e-4 relative plus a small absolute
# term for coefficients that cancel to near zero.
RTOL, ATOL = 1e-4, 1e-4


def close(a, b):
    return np.all(np.abs(a - b) <= RTOL * np.abs(b) + ATOL)


def configs():
    return [dict(n_ceps=16, filter_width=268.258), dict(n_ceps=40, filter_width=138.0)]


@pytest.mark.parametrize("cfg", configs())
def test_cfg1_ten_seconds(ctx, cfg):
    import rasr_amd
    from oracle import OracleMfcc
    pcm = synth.waveform(160000, seed=1)
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=cfg["n_ceps"], filter_width=cfg["filter_width"])
    got = fe.run(pcm)
    want = OracleMfcc(**cfg).run(pcm)
    assert got.shape == want.shape == (999, cfg["n_ceps"])
    assert close(got, want), np.abs(got - want).max()
    # tight check on the bulk: median relative error is at f32 round-off level
    assert np.median(np.abs(got - want) / (np.abs(want) + 1e-3)) < 2e-6


@pytest.mark.parametrize("n", [1, 2, 159, 160, 399, 400, 401, 560, 561, 800, 801, 1000, 4096, 5281, 48077])
def test_ragged_lengths_and_short_last_frame(ctx, n):
    import rasr_amd
    from oracle import OracleMfcc
    pcm = synth.waveform(n, seed=100 + n)
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=16)
    got = fe.run(pcm)
    want = OracleMfcc(n_ceps=16).run(pcm)
    assert got.shape == want.shape
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)      # log10(0) = -inf propagates like the reference
    assert close(got[fin], want[fin]), np.abs(got[fin] - want[fin]).max()


def test_empty_segment(ctx):
    import rasr_amd
    fe = rasr_amd.MfccExtractor(ctx)
    assert fe.run(np.zeros(0, np.float32)).shape == (0, 16)


def test_batch_equals_single(ctx):
    import rasr_amd
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=40, filter_width=138.0)
    lens = [16000, 1, 401, 33333, 0, 8000]
    pcms = [synth.waveform(n, seed=7 + i) for i, n in enumerate(lens)]
    outs = fe.run_batch(pcms)
    for p, o in zip(pcms, outs):
        single = fe.run(p)
        assert np.array_equal(single.view(np.uint32), o.view(np.uint32))


@pytest.mark.parametrize("alpha", [0.97, 0.0])
def test_preemphasis_alpha(ctx, alpha):
    import rasr_amd
    from oracle import OracleMfcc
    pcm = synth.waveform(20000, seed=5)
    got = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=16, alpha=alpha).run(pcm)
    want = OracleMfcc(n_ceps=16, alpha=alpha).run(pcm)
    assert close(got, want), np.abs(got - want).max()


@pytest.mark.parametrize("fs,width", [(8000.0, 268.258), (11025.0, 200.0), (22050.0, 268.258), (32000.0, 268.258), (44100.0, 268.258)])
def test_other_sample_rates(ctx, fs, width):
    """FFT lengths 256 .. 2048 (radix-4 plus the radix-2 tail stage)."""
    import rasr_amd
    from oracle import OracleMfcc
    pcm = synth.waveform(int(fs * 1.3), seed=11, fs=fs)
    got = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=12, sample_rate=fs, filter_width=width).run(pcm)
    want = OracleMfcc(n_ceps=12, sample_rate=fs, filter_width=width).run(pcm)
    assert got.shape == want.shape
    assert close(got, want), np.abs(got - want).max()


def test_device_resident_plan_matches_host_path(ctx):
    import torch

    import rasr_amd
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=40, filter_width=138.0)
    lens = synth.utterance_lengths(12, seed=3, lo_s=0.5, hi_s=2.0)
    pcms = [synth.waveform(int(n), seed=50 + i) for i, n in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum(lens)])
    plan = fe.plan(off)
    pcm_dev = torch.from_numpy(np.concatenate(pcms)).cuda()
    ceps_dev = torch.empty((plan.total_frames, 40), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    fe.run_plan(plan, pcm_dev, ceps_dev)
    torch.cuda.synchronize()
    got = ceps_dev.cpu().numpy()
    outs = fe.run_batch(pcms)
    for u in range(len(pcms)):
        seg = got[plan.frame_offsets[u]:plan.frame_offsets[u + 1]]
        assert np.array_equal(seg.view(np.uint32), outs[u].view(np.uint32))


def test_full_size_properties(ctx):
    """BASELINE config 2 scale (1000 utterances): size-independent checks -- frame counts, finiteness,
    and equality of every 97th utterance with the oracle."""
    import torch

    import rasr_amd
    from oracle import OracleMfcc
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=40, filter_width=138.0)
    orc = OracleMfcc(n_ceps=40, filter_width=138.0)
    lens = synth.utterance_lengths(1000, seed=3)
    off = np.concatenate([[0], np.cumsum(lens)])
    base = synth.waveform(int(lens.max()), seed=4)
    # utterance u = base waveform rotated by u samples (cheap to build, all different)
    pcm = np.concatenate([np.roll(base, u)[:n] for u, n in enumerate(lens)])
    plan = fe.plan(off)
    assert plan.total_frames == sum(orc.n_frames(int(n)) for n in lens)
    pcm_dev = torch.from_numpy(pcm).cuda()
    ceps_dev = torch.empty((plan.total_frames, 40), dtype=torch.float32, device="cuda")
    ctx.use_torch_stream()
    fe.run_plan(plan, pcm_dev, ceps_dev)
    torch.cuda.synchronize()
    got = ceps_dev.cpu().numpy()
    assert np.isfinite(got).all()
    for u in range(0, 1000, 97):
        want = orc.run(pcm[off[u]:off[u + 1]])
        seg = got[plan.frame_offsets[u]:plan.frame_offsets[u + 1]]
        assert close(seg, want), (u, np.abs(seg - want).max())


# MF-PLP (mfplp.flow): the chain adds the power node (^0.33), an autocorrelation transform, the Levinson recursion (f64) and the LPC
# cepstrum recursion (f32) behind the filter bank.  The power node is the reference's ::pow(double, double) narrowed to f32 on both sides
# (round 2; the earlier v_exp / v_log based __powf was ~10 ulp off and the recursions amplified that to 2e-3), so what is left is the
# FFT's rounding carried through the recursions: the MFCC band holds (fuzz campaign: worst 1.1e-5 relative).
PLP_RTOL, PLP_ATOL = 1e-4, 1e-4


@pytest.mark.parametrize("nc,nac", [(13, 13), (9, 20), (2, 2)])
def test_mfplp_ten_seconds(ctx, nc, nac):
    import rasr_amd
    from oracle import OracleMfcc
    from oracle.binding import MfccCfg
    pcm = synth.waveform(160000, seed=3)
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=nc, front_end="mfplp", nr_autocorrelation_coefficients=nac, normalize=True)
    got = fe.run(pcm)
    want = OracleMfcc(MfccCfg.mfplp(n_ceps=nc, n_autocorrelation=nac)).run(pcm)
    assert got.shape == want.shape == (999, nc) and np.all(np.isfinite(got))
    assert np.all(np.abs(got - want) <= PLP_RTOL * np.abs(want) + PLP_ATOL), np.abs(got - want).max()
    assert np.median(np.abs(got - want) / (np.abs(want) + 1e-2)) < 2e-5


def test_mfplp_ragged_batch_silence_and_device_plan(ctx):
    """segments of different lengths in one batch equal the single calls; digital silence makes the Levinson recursion fail:
    NaN frames exactly where the oracle has them (the reference reports an error for such frames); the device-resident plan
    entry point writes the same features"""
    import torch

    import rasr_amd
    from oracle import OracleMfcc
    from oracle.binding import MfccCfg
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=12, front_end="mfplp", nr_autocorrelation_coefficients=16, normalize=True)
    o = OracleMfcc(MfccCfg.mfplp(n_ceps=12, n_autocorrelation=16))
    segs = [synth.waveform(n, seed=200 + n) for n in (401, 5281, 160, 48077)]
    segs.append(np.concatenate([synth.waveform(2000, seed=7), np.zeros(3000, np.float32), synth.waveform(1500, seed=8)]))
    outs = fe.run_batch(segs)
    for x, got in zip(segs, outs):
        want = o.run(x)
        assert got.shape == want.shape
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(got), nan)
        assert np.all(np.abs(got[~nan] - want[~nan]) <= PLP_RTOL * np.abs(want[~nan]) + PLP_ATOL)
        assert np.array_equal(fe.run(x), got, equal_nan=True)
    assert np.isnan(outs[-1]).any() and not np.isnan(outs[0]).any()
    ctx.use_torch_stream()
    off = np.concatenate([[0], np.cumsum([len(x) for x in segs])])
    plan = fe.plan(off)
    pcm = torch.from_numpy(np.concatenate(segs)).cuda()
    ceps = torch.empty((plan.total_frames, 12), dtype=torch.float32, device="cuda")
    fe.run_plan(plan, pcm, ceps)
    torch.cuda.synchronize()
    assert np.array_equal(ceps.cpu().numpy(), np.concatenate(outs), equal_nan=True)


# PLP (plp.flow): bark / trapeze / include-boundary filter bank, duplicated first / last output, equal-loudness weighting, then the
# MF-PLP tail; same tolerance band as MF-PLP
@pytest.mark.parametrize("fs,spacing,nc,nac", [(16000.0, 0.93853, 13, 13), (16000.0, 0.93853, 9, 20), (8000.0, 0.973442, 11, 11)])
def test_plp_ten_seconds(ctx, fs, spacing, nc, nac):
    import rasr_amd
    from oracle import OracleMfcc
    from oracle.binding import MfccCfg
    n = int(10 * fs)
    pcm = synth.waveform(n, seed=5)
    fe = rasr_amd.MfccExtractor.plp(ctx, nr_cepstrum_coefficients=nc, nr_autocorrelation_coefficients=nac, sample_rate=fs, spacing=spacing)
    o = OracleMfcc(MfccCfg.plp(n_ceps=nc, n_autocorrelation=nac, spacing=spacing, sample_rate=fs))
    assert (fe.n_filters, fe.info.n_transform_inputs) == (o.n_filters, o.n_filters + 2)
    assert np.array_equal(fe.equal_loudness(), o.equal_loudness)
    got, want = fe.run(pcm), o.run(pcm)
    assert got.shape == want.shape == (999, nc) and np.all(np.isfinite(got))    # 20 ms window: (n - 320) / 160 + 1 frames at 16 kHz
    assert np.all(np.abs(got - want) <= PLP_RTOL * np.abs(want) + PLP_ATOL), np.abs(got - want).max()
    assert np.median(np.abs(got - want) / (np.abs(want) + 1e-2)) < 2e-5


def test_plp_golden_and_ragged_batch(ctx):
    """the committed oracle outputs (tests/golden/orc_plp.npz) and a ragged batch incl. a one-sample and a silent segment"""
    import rasr_amd
    from oracle import OracleMfcc
    from oracle.binding import MfccCfg
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "orc_plp.npz"))
    pcm = synth.waveform(16000, seed=1)
    fe = rasr_amd.MfccExtractor.plp(ctx)
    got = fe.run(pcm)
    assert np.all(np.abs(got - g["plp16k"]) <= PLP_RTOL * np.abs(g["plp16k"]) + PLP_ATOL)
    o = OracleMfcc(MfccCfg.plp())
    segs = [synth.waveform(n, seed=300 + n) for n in (1, 319, 320, 321, 5281, 48077)]
    segs.append(np.concatenate([synth.waveform(2000, seed=7), np.zeros(3000, np.float32), synth.waveform(1500, seed=8)]))
    outs = fe.run_batch(segs)
    for x, y in zip(segs, outs):
        want = o.run(x)
        assert y.shape == want.shape
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(y), nan)
        assert np.all(np.abs(y[~nan] - want[~nan]) <= PLP_RTOL * np.abs(want[~nan]) + PLP_ATOL)
    assert np.isnan(outs[-1]).any()


@pytest.mark.parametrize("kw,okw", [
    (dict(type="trapeze", boundary="include-boundary", warping_function="bark", filter_width=3.8, spacing=0.9),
     dict(filter_type=1, boundary=1, warping=1, mel_filter_width=3.8, mel_spacing=0.9)),
    (dict(boundary="emphasize-boundary", filter_width=300.0), dict(boundary=2, mel_filter_width=300.0)),
    (dict(warping_function="bark", filter_width=2.0), dict(warping=1, mel_filter_width=2.0))])
def test_mfcc_with_other_filter_banks(ctx, kw, okw):
    """signal-filterbank's other types / boundaries / warping functions in front of the MFCC tail (log10 + DCT)"""
    import rasr_amd
    from oracle import OracleMfcc
    from oracle.binding import MfccCfg
    cfg = MfccCfg.default(n_ceps=12)
    for k, v in okw.items():
        setattr(cfg, k, v)
    pcm = synth.waveform(48000, seed=11)
    got = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=12, **kw).run(pcm)
    want = OracleMfcc(cfg).run(pcm)
    assert got.shape == want.shape and close(got, want), np.abs(got - want).max()


def test_lpc_cepstrum_register_and_lds_kernels_agree(ctx, monkeypatch):
    """the register-resident LPC / cepstrum kernel (<= 24 autocorrelation coefficients) and the LDS kernel run the same operations in
    the same order: identical bits, NaN frames included; more than 24 coefficients take the LDS kernel and match the oracle"""
    import rasr_amd
    from oracle import OracleMfcc
    from oracle.binding import MfccCfg
    pcm = np.concatenate([synth.waveform(30000, seed=21), np.zeros(2000, np.float32), synth.waveform(9000, seed=22)])
    for nac, nc in ((13, 13), (16, 12), (17, 17), (24, 20)):
        a = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=nc, front_end="mfplp", nr_autocorrelation_coefficients=nac, normalize=True,
                                   filter_width=138.0, tuning="lpc=regs").run(pcm)
        b = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=nc, front_end="mfplp", nr_autocorrelation_coefficients=nac, normalize=True,
                                   filter_width=138.0, tuning="lpc=lds").run(pcm)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.isnan(a).any()
    got = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=25, front_end="mfplp", nr_autocorrelation_coefficients=30, normalize=True,
                                 filter_width=138.0).run(pcm)
    cfg = MfccCfg.mfplp(n_ceps=25, n_autocorrelation=30, filter_width=138.0)
    want = OracleMfcc(cfg).run(pcm)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.median(np.abs(got[~nan] - want[~nan]) / (np.abs(want[~nan]) + 1e-2)) < 1e-4      # order-29 recursions: ill-conditioned tail


def test_s16_samples_give_the_same_bits_as_f32(ctx):
    """amx_mfcc_run_s16 / _run_batch_s16 / _run_plan_dev_s16: the samples as the audio file holds them (Flow/TypeConverter.hh:35-43 widens
    s16 to f32 without scaling in front of the chain; here inside the kernel) -- bit-identical cepstra, ragged batch, every front end"""
    import torch

    import rasr_amd
    for kw in (dict(nr_cepstrum_coefficients=40, filter_width=138.0), dict(nr_cepstrum_coefficients=16),
               dict(nr_cepstrum_coefficients=16, front_end="mfplp", nr_autocorrelation_coefficients=20, normalize=True)):
        fe = rasr_amd.MfccExtractor(ctx, **kw)
        lens = [16000, 401, 399, 7, 0, 52345]
        pcm = [synth.waveform(n, seed=70 + i) for i, n in enumerate(lens)]
        assert all(np.array_equal(p, np.rint(p)) and np.abs(p).max(initial=0) <= 32767 for p in pcm)
        want = fe.run_batch(pcm)
        got = fe.run_batch([p.astype(np.int16) for p in pcm])
        for w, g in zip(want, got):
            assert w.shape == g.shape and np.array_equal(w.view(np.uint32), g.view(np.uint32))
        one = fe.run(pcm[0].astype(np.int16))
        assert np.array_equal(one.view(np.uint32), want[0].view(np.uint32))
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        plan = fe.plan(off)
        cat = np.concatenate(pcm)
        ctx.use_torch_stream()
        out32 = torch.empty((plan.total_frames, fe.n_ceps), dtype=torch.float32, device="cuda")
        out16 = torch.empty_like(out32)
        fe.run_plan(plan, torch.from_numpy(cat).cuda(), out32)
        fe.run_plan(plan, torch.from_numpy(cat.astype(np.int16)).cuda(), out16)
        torch.cuda.synchronize()
        assert torch.equal(out32.view(torch.int32), out16.view(torch.int32))
        assert np.array_equal(out32.cpu().numpy().view(np.uint32), np.concatenate(want).view(np.uint32))


@pytest.mark.parametrize("fft", ["mfma", "r16"])
def test_matrix_core_fft_against_the_butterfly_fft_and_the_oracle(ctx, tmp_path, fft):
    """amx_mfcc_cfg.tuning "fft=r16": radix-16 register butterflies, four frames per wave, two LDS round trips per transform.
    amx_mfcc_cfg.tuning "fft=mfma" runs the 512-point transform as two 16x16x16 complex products on v_mfma_f32_16x16x4_f32 (slower than the
    radix-4 LDS stages, kept for A/B runs): within the MFCC bar of the oracle and within f32 round-off of the default kernel --
    incl. the transform's corner cases: a unit impulse at every position class, a constant, a tone on a bin, silence"""
    import rasr_amd
    from oracle import OracleMfcc
    fe = rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=40, filter_width=138.0)
    orc = OracleMfcc(n_ceps=40, filter_width=138.0)
    n = 4000
    t = np.arange(n)
    sigs = [synth.waveform(n, seed=91), np.full(n, 1000.0, np.float32), (8000 * np.sin(2 * np.pi * t * 1000 / 16000)).astype(np.float32),
            np.zeros(n, np.float32)]
    for pos in (0, 1, 63, 64, 199, 200, 399, 400, 1234):
        imp = np.zeros(n, np.float32)
        imp[pos] = 20000.0
        sigs.append(imp)
    got = np.stack(rasr_amd.MfccExtractor(ctx, nr_cepstrum_coefficients=40, filter_width=138.0, tuning="fft=" + fft).run_batch(sigs))
    ref = np.stack(fe.run_batch(sigs))
    for g, r, x in zip(got, ref, sigs):
        want = orc.run(x)
        ok = np.isfinite(want)
        assert np.array_equal(np.isfinite(g), ok)
        assert np.all(np.abs(g[ok] - want[ok]) <= 1e-4 * np.abs(want[ok]) + 1e-4), np.abs(g[ok] - want[ok]).max()
        assert np.all(np.abs(g[ok] - r[ok]) <= 1e-4 * np.abs(r[ok]) + 1e-4)


@pytest.mark.parametrize("alpha", [1.0, 0.97, 0.0])
def test_radix16_fft_ragged_batch_s16_and_preemphasis(ctx, alpha):
    """fft=r16 on a ragged batch (segments shorter than a frame, frame counts off the 4-frame wave group and the 16-frame tile, first
    and last frames of a segment through the guarded loads), every pre-emphasis form, f32 and s16 samples: within the MFCC bar of
    the oracle; s16 and f32 samples of whole-numbered audio give the same bits"""
    import rasr_amd
    from oracle import OracleMfcc
    lens = [100, 399, 400, 401, 560, 561, 720, 1040, 1360, 2960, 3000, 16000, 16001, 33333]
    sigs = [np.round(synth.waveform(n, seed=300 + i)).astype(np.float32) for i, n in enumerate(lens)]
    kw = dict(nr_cepstrum_coefficients=40, filter_width=138.0, alpha=alpha)
    fe = rasr_amd.MfccExtractor(ctx, tuning="fft=r16", **kw)
    orc = OracleMfcc(n_ceps=40, filter_width=138.0, alpha=alpha)
    got = fe.run_batch(sigs)
    for g, x in zip(got, sigs):
        want = orc.run(x)
        assert g.shape == want.shape, (g.shape, want.shape, len(x))
        ok = np.isfinite(want)
        assert np.array_equal(np.isfinite(g), ok)
        assert np.all(np.abs(g[ok] - want[ok]) <= 1e-4 * np.abs(want[ok]) + 1e-4), (len(x), np.abs(g[ok] - want[ok]).max())
    got16 = fe.run_batch([x.astype(np.int16) for x in sigs])
    for g, g16 in zip(got, got16):
        assert g.shape == g16.shape and np.array_equal(g.view(np.uint32), g16.view(np.uint32))


@pytest.mark.parametrize("kw", [dict(nr_cepstrum_coefficients=40, filter_width=138.0), dict(nr_cepstrum_coefficients=16, alpha=0.97),
                                dict(nr_cepstrum_coefficients=16, front_end="mfplp", nr_autocorrelation_coefficients=20, normalize=True)])
def test_sample_prefetch_variant_gives_the_same_bits(ctx, kw):
    """tuning prefetch=1 | 0: a wave fetches its next frame's samples (within the tile, and across to the workgroup's next tile) while it
    transforms the current one -- the load path differs, the arithmetic does not: bit-identical cepstra on a ragged batch whose
    segments end inside, at and just behind the 512-sample span of a frame (the frames that may not use the unguarded loads), f32 and
    s16 samples, more tiles than workgroups"""
    import rasr_amd
    lens = [0, 7, 399, 400, 401, 511, 512, 513, 560, 672, 673, 1000, 2960, 16000, 16001, 33333] + [4000 + 37 * i for i in range(40)]
    sigs = [np.round(synth.waveform(n, seed=500 + i)).astype(np.float32) for i, n in enumerate(lens)]
    big = [np.round(synth.waveform(160 * 3000 + 240, seed=499)).astype(np.float32)] * 6     # 6 x 3000 frames: every workgroup walks several tiles
    on = rasr_amd.MfccExtractor(ctx, tuning="prefetch=1", **kw)
    off = rasr_amd.MfccExtractor(ctx, tuning="prefetch=0", **kw)
    for batch in (sigs, big, sigs + big):
        a, b = on.run_batch(batch), off.run_batch(batch)
        for x, y in zip(a, b):
            assert x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32))
        a16 = on.run_batch([x.astype(np.int16) for x in batch])
        for x, y in zip(a16, b):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
