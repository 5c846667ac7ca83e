"""tools/fuzz_frontends.py [n_cases] [seed] -- random configurations of the round-2 front ends against the oracle (GPU box):
signal-filterbank types / boundaries / warpings in front of the MFCC, MF-PLP and PLP tails (sample rates, window and shift lengths,
filter widths and spacings, cepstrum / autocorrelation orders, ragged segment lengths down to one sample), the gammatone chain
(channels, cascade, centre-frequency modes, warping, windows, spectral integration, root compression, cosine transform) and the
per-vector normalisers on strided views.  A configuration one side rejects must be rejected by the other side as well.
Companion of tools/fuzz_more.py; prints MISMATCH lines and a summary, exit code 1 on any mismatch."""
import os
import sys

import numpy as np
import torch  # before the library touches HIP

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rasr_amd  # noqa: E402
from oracle import OracleMfcc  # noqa: E402
from oracle.binding import GammatoneCfg, MfccCfg, OracleGammatone, oracle_vector_normalize  # noqa: E402
from tests import synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 1))
ctx = rasr_amd.Context(0)
bad, ran, rejected, conditioned = 0, {"mfcc": 0, "mfplp": 0, "plp": 0, "gammatone": 0, "vnorm": 0}, 0, 0
worst = {"mfcc": 0.0, "mfplp": 0.0, "plp": 0.0}   # max |got - want| / (|want| + 1) seen per front end


def fail(what, **kw):
    global bad
    bad += 1
    print("MISMATCH", what, kw, flush=True)


def lengths(fs):
    base = [1, 2, int(rng.integers(3, 400)), int(rng.integers(400, 5000)), int(rng.integers(5000, 3 * fs))]
    return [base[i] for i in rng.choice(len(base), 3, replace=False)]


for case in range(n_cases):
    seed = int(rng.integers(1, 1 << 30))
    ctx.set_contract("off")   # the filter-bank front ends are checked against the contract=off oracle (tables bit for bit, cepstra at the bar)
    # ------------------------------------------------------------------ filter-bank front ends
    fs = float(rng.choice([8000.0, 11025.0, 16000.0, 22050.0, 44100.0]))
    fe_name = str(rng.choice(["mfcc", "mfplp", "plp"]))
    warping = str(rng.choice(["mel", "bark"])) if fe_name != "plp" else "bark"
    ftype = str(rng.choice(["triangular", "trapeze"]))
    boundary = str(rng.choice(["stretch-to-cover", "include-boundary", "emphasize-boundary"]))
    width = float(rng.uniform(120, 400)) if warping == "mel" else float(rng.uniform(1.5, 5.0))
    spacing = 0.0 if rng.integers(0, 2) else (float(rng.uniform(60, 200)) if warping == "mel" else float(rng.uniform(0.7, 1.2)))
    length = float(rng.choice([0.01, 0.016, 0.02, 0.025, 0.032]))
    shift = float(rng.choice([0.005, 0.01, 0.0125]))
    alpha = float(rng.choice([0.0, 0.95, 1.0]))
    nc = int(rng.integers(1, 21))
    nac = int(rng.integers(2, 26)) if fe_name != "mfcc" else 0
    if fe_name != "mfcc" and rng.integers(0, 8):   # mostly orders the chain accepts (2 <= cepstra <= autocorrelations <= filter outputs)
        try:
            nf = OracleMfcc(MfccCfg(fs, length, shift, alpha, length, 1, width, spacing, 1, 1, 0, 0, 0, 0.33, {"triangular": 0, "trapeze": 1}[ftype],
                                    {"stretch-to-cover": 0, "include-boundary": 1, "emphasize-boundary": 2}[boundary],
                                    {"mel": 0, "bark": 1}[warping])).n_filters + (2 if fe_name == "plp" else 0)
            if nf >= 2:
                nac = int(rng.integers(2, min(nf, 26) + 1))
                nc = int(rng.integers(2, nac + 1))
        except Exception:
            pass
    kw = dict(nr_cepstrum_coefficients=nc, filter_width=width, sample_rate=fs, alpha=alpha, length=length, shift=shift,
              maximum_input_size=length, spacing=spacing, normalize=fe_name != "mfcc", front_end=fe_name,
              nr_autocorrelation_coefficients=nac, type=ftype, boundary=boundary, warping_function=warping)
    cfg = MfccCfg(fs, length, shift, alpha, length, 1, width, spacing, 1, nc, int(fe_name != "mfcc"), {"mfcc": 0, "mfplp": 1, "plp": 2}[fe_name], nac,
                  0.33, {"triangular": 0, "trapeze": 1}[ftype], {"stretch-to-cover": 0, "include-boundary": 1, "emphasize-boundary": 2}[boundary],
                  {"mel": 0, "bark": 1}[warping])
    o = fe = None
    try:
        o = OracleMfcc(cfg)
    except Exception:
        pass
    try:
        fe = rasr_amd.MfccExtractor(ctx, **kw)
    except rasr_amd.AmxError:
        pass
    if (o is None) != (fe is None):
        fail("front-end accept/reject", oracle=o is not None, product=fe is not None, kw=kw)
    elif o is None:
        rejected += 1
    else:
        ran[fe_name] += 1
        if (fe.n_filters, fe.frame_len, fe.frame_shift, fe.fft_len) != (o.n_filters, o.frame_len, o.frame_shift, o.fft_len):
            fail("front-end geometry", kw=kw)
        else:
            segs = [synth.waveform(n, seed=seed + i) * np.float32(rng.choice([1.0, 0.01, 30.0])) for i, n in enumerate(lengths(int(fs)))]
            outs = fe.run_batch(segs)
            rt, at = 1e-4, 1e-4
            for x, y in zip(segs, outs):
                want = o.run(x)
                if y.shape != want.shape:
                    fail("front-end shape", kw=kw, n=len(x), got=y.shape, want=want.shape)
                    continue
                fin = np.isfinite(want)
                if not np.array_equal(np.isfinite(y), fin) or not np.array_equal(y[~fin], want[~fin], equal_nan=True):
                    # MF-PLP / PLP: a recursion that overflows on one side only is a conditioning effect of a degenerate frame, not a defect
                    if fe_name == "mfcc" or np.mean(np.isfinite(y) != fin) > 0.01:
                        fail("front-end non-finite pattern", kw=kw, n=len(x), differ=int(np.sum(np.isfinite(y) != fin)))
                    continue
                if fin.any():
                    worst[fe_name] = max(worst[fe_name], float(np.max(np.abs(y[fin] - want[fin]) / (np.abs(want[fin]) + 1.0))))
                err = np.abs(y[fin] - want[fin]) - rt * np.abs(want[fin])
                frac_bad = float(np.mean(err > at)) if err.size else 0.0
                # the LPC recursions amplify the device pow's ulps by the conditioning of the autocorrelation matrix: a band for
                # MF-PLP / PLP (a handful of ill-conditioned frames may leave it), a hard bound for MFCC
                if (fe_name != "mfcc" and frac_bad > 0.002) or (fe_name == "mfcc" and frac_bad > 0):
                    # conditioning guard: how far does the ORACLE itself move when every sample moves by 2e-6 relative (random signs)?
                    # That is the size of the device's spectrum deviation -- f32 FFT with table twiddles against the reference's
                    # f64-recurrence twiddles, <= 2e-5 on MFCC cepstra.  A frame whose Levinson recursion (LPC order 24 from 39 smooth
                    # mel outputs, a 2-sample segment under a 441-sample window ...) turns that into more than the bar is
                    # ill-conditioned, not wrong -- tools/dbg_mfplp.py: the oracle moves by up to 65 bars under ONE ulp there -- and
                    # the bar widens by five times the oracle's own movement.  Well-conditioned frames move by ~1e-5: nothing is masked.
                    # MFCC too: a mel filter over bins 100 dB below the spectrum's peak (44.1 kHz audio, a tone + noise) carries the FFT's
                    # absolute rounding error -- the reference's f32-data FFT has the same class of error -- as a large RELATIVE one, and
                    # log10 hands it on (seed 7: one cepstrum of 1840 off by 2.1e-4).
                    prng = np.random.Generator(np.random.PCG64(seed))
                    sens = np.zeros(int(fin.sum()))
                    for _ in range(6):   # the movement is directional: the largest of six sign patterns, and per frame the largest over its cepstra
                        prt = prng.choice(np.array([-2e-6, 2e-6], np.float32), size=x.shape)
                        want2 = o.run((x * (np.float32(1.0) + prt)).astype(np.float32))
                        if want2.shape != want.shape:
                            continue
                        d = np.abs(want2 - want)
                        d = np.where(np.isfinite(d), d, np.inf)
                        d = np.broadcast_to(d.max(axis=1, keepdims=True), d.shape)
                        sens = np.maximum(sens, d[fin])
                    if True:
                        err = err - 5.0 * sens
                        frac_bad = float(np.mean(err > at))
                        conditioned += 1
                if (fe_name == "mfcc" and frac_bad > 0) or frac_bad > 0.002:
                    fail("front-end values", kw=kw, n=len(x), frac=frac_bad, worst=float(err.max()))
    # ------------------------------------------------------------------ gammatone
    gfs = float(rng.choice([8000.0, 16000.0]))
    gkw = dict(sample_rate=gfs, cascade=int(rng.integers(1, 7)), channels=int(rng.integers(4, 90)), cf_mode=int(rng.integers(0, 2)),
               min_freq=float(rng.uniform(50, 300)), max_freq=float(rng.uniform(0.3, 0.49)) * gfs,
               warping_factor=float(rng.choice([1.0, 1.0, 0.9, 1.1])), warp_freq_break=float(rng.uniform(0.3, 0.45)) * gfs,
               ti_window=int(rng.integers(0, 2)), ti_length_s=float(rng.choice([0.01, 0.02, 0.025, 0.032])),
               ti_shift_s=float(rng.choice([0.004, 0.01, 0.016])))
    ctx.set_contract("off")          # (the MFCC-family handles above and the gammatone handle below name their arithmetic themselves)
    mode = int(rng.integers(0, 4))   # 0 temporal integration only; 1 + spectral; 2 + root; 3 + cosine transform
    if mode >= 1:
        gkw.update(si_length=int(rng.integers(1, 12)), si_shift=int(rng.integers(1, 6)), si_window=int(rng.integers(0, 2)))
    if mode >= 2:
        gkw.update(power=float(rng.choice([0.1, 0.33, 0.5])))
    if mode >= 3:
        gkw.update(n_ceps=int(rng.integers(1, 16)), dct_normalize=int(rng.integers(0, 2)))
    o = fe = None
    gcontract = ("off", "fma")[int(rng.integers(0, 2))]   # the reference's two arithmetics, through the handle's own tuning string
    try:
        o = OracleGammatone(GammatoneCfg.default(**gkw), contract=gcontract)
    except Exception:
        pass
    try:
        fe = rasr_amd.GammatoneExtractor(ctx, tuning="contract=" + gcontract, **gkw)
    except rasr_amd.AmxError:
        pass
    if (o is None) != (fe is None):
        fail("gammatone accept/reject", oracle=o is not None, product=fe is not None, kw=gkw)
    elif o is None:
        rejected += 1
    else:
        ran["gammatone"] += 1
        cf, co = fe.tables()
        if not (np.array_equal(cf.view(np.uint32), o.center_frequencies.view(np.uint32)) and np.array_equal(co.view(np.uint32), o.coefficients.view(np.uint32))):
            fail("gammatone tables", kw=gkw)
        for n in lengths(int(gfs)):
            x = synth.waveform(min(n, 20000), seed=seed + n)
            want, got = o.run(x), fe.run(x)
            if got.shape != want.shape:
                fail("gammatone shape", kw=gkw, n=len(x), got=got.shape, want=want.shape)
            elif not np.array_equal(np.isnan(got), np.isnan(want)) or not np.array_equal(got[~np.isnan(want)].view(np.uint32), want[~np.isnan(want)].view(np.uint32)):
                # every stage is IEEE arithmetic in a fixed order; the root compression is the node's f64 pow narrowed to f32 (a negative
                # first sample under a rectangular window gives NaN on both sides: same places, the payload bits are not compared)
                fail("gammatone bits", kw=gkw, n=len(x), err=float(np.nanmax(np.abs(got - want))))
    # ------------------------------------------------------------------ per-vector normalisers
    ctx.use_torch_stream()
    n, dim = int(rng.integers(1, 3000)), int(rng.integers(2, 300))
    pad_in, pad_out = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    x = (rng.standard_normal((n, dim)) * rng.choice([1e-3, 1.0, 1e3])).astype(np.float32)
    if rng.integers(0, 3) == 0:
        x[int(rng.integers(0, n))] = 0.0
    wide = torch.zeros((n, dim + pad_in), dtype=torch.float32, device="cuda")
    wide[:, :dim] = torch.from_numpy(x).cuda()
    vcontract = ("off", "fma")[int(rng.integers(0, 2))]
    ctx.set_contract(vcontract)
    for kind in rasr_amd.Context.VECTOR_NORMALIZATIONS:
        ran["vnorm"] += 1
        out = torch.full((n, dim + pad_out), 7.0, dtype=torch.float32, device="cuda")
        ctx.vector_normalize(kind, wide, dim + pad_in, n, dim, out, dim + pad_out)
        torch.cuda.synchronize()
        got, want = out[:, :dim].cpu().numpy(), oracle_vector_normalize(x, kind, contract=vcontract)
        if not np.array_equal(got.view(np.uint32), want.view(np.uint32)) or not bool((out[:, dim:] == 7.0).all()):
            fail("vector normalisation", kind=kind, n=n, dim=dim)

print("fuzz_frontends: %d cases, ran %s, %d configurations rejected by both sides, %d segments judged with the conditioning guard, %d mismatches; worst relative deviation %s" % (n_cases, ran, rejected, conditioned, bad, worst))
sys.exit(1 if bad else 0)
