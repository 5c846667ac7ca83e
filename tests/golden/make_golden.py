"""Regenerates the golden vectors in tests/golden/.  Run in the build container only:

    python tests/golden/make_golden.py

* ref_*.npz / ref_*.json are OUTPUTS OF THE REFERENCE ITSELF: oracle/_ref/libref.so is built by
  oracle/ref/Makefile from unmodified reference translation units (Math::FastFourierTransform,
  Signal::WindowBuffer, the mel warping functors, Mm::gaussLogNormFactor / inverseSquareRoot,
  Math::Matrix<f32> * Math::Vector<f32>, Math::transformAlternatingComplex / pointerAbs, the bark warping functors and
  Math::EqualLoudnessPreemphasis[4Khz]).
* survey_c1.json holds the known answers the reference produced in this container during the survey
  (SURVEY.md Appendix C.1).
* nn_kat.json is transcribed from the reference's unit tests (see its "source" field).
* orc_*.npz are outputs of the oracle (oracle/liboracle.so) AFTER it has been pinned against all of the
  above; they let the GPU box (which has no reference tree) detect any drift of the oracle or the kernels.
Only inputs and expected outputs are stored -- no reference source text.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import OracleGmm, OracleMfcc, load_ref, oracle_ffnn_score  # noqa: E402
from tests import synth  # noqa: E402


def bark_golden(R):
    """ref_bark.json: bark warping (value / derivative / inverse, alone and nested with disc-to-cont), both equal-loudness curves and
    plp.flow's composed f(index), evaluated by the reference's own analytic-function classes (f64 as hex strings); plus
    orc_plp.npz, outputs of the oracle's plp.flow chain after it has been pinned on those"""
    rows = []
    for f in [0.0, 31.25, 62.5, 440.0, 1234.5, 3999.99, 4000.0, 7968.75, 8000.0]:
        b = R.ref_bark(f)
        rows.append(dict(f=float(f).hex(), bark=float(b).hex(), dbark=float(R.ref_bark_derivative(f)).hex(),
                         inv=float(R.ref_bark_inverse(b)).hex(), eql=float(R.ref_equal_loudness(f, 0)).hex(),
                         eql4k=float(R.ref_equal_loudness(f, 1)).hex()))
    bins = []
    for sr in (0.032, 0.064):
        for b in [0, 1, 7, 100, 255, 256]:
            w = R.ref_bark_bin(b, sr)
            bins.append(dict(bin=b, sr=sr, warped=float(w).hex(), dwarped=float(R.ref_bark_bin_derivative(b, sr)).hex(),
                             back=float(R.ref_bark_bin_inverse(w, sr)).hex()))
    idx = []
    for sr_text, n, four in (("1.0655", 22, 0), ("1.02728", 17, 1)):   # "%g" of 1 / 0.93853 and of 1 / 0.973442 (plp.flow's two spacings)
        sr = float(sr_text)
        idx.append(dict(sr=sr, four_khz=four, values=[float(R.ref_plp_equal_loudness(float(i), sr, four)).hex() for i in range(n)]))
    json.dump(dict(bark=rows, bins=bins, plp_f=idx), open(os.path.join(HERE, "ref_bark.json"), "w"), indent=0)
    from oracle.binding import MfccCfg
    pcm = synth.waveform(16000, seed=1)
    gold = {}
    for tag, cfg in (("plp16k", MfccCfg.plp(n_ceps=13, n_autocorrelation=13)),
                     ("plp8k", MfccCfg.plp(n_ceps=11, n_autocorrelation=11, spacing=0.973442, sample_rate=8000.0))):
        m = OracleMfcc(cfg)
        gold[tag] = m.run(pcm)
        for k, v in m.stages(pcm, 3).items():
            if k in ("mel", "logmel", "ceps"):
                gold[tag + "_f3_" + k] = v
        gold[tag + "_eql"] = m.equal_loudness
        s, e, o, w = m.filters
        gold[tag + "_fstart"], gold[tag + "_fend"], gold[tag + "_fweights"] = s, e, w
    np.savez_compressed(os.path.join(HERE, "orc_plp.npz"), **gold)


def time_framing_golden(R):
    """ref_time_framing.json: frames of Signal::TimeWindowBuffer<Flow::Vector<f32>> (the base of signal-temporalintegration) driven
    like SlidingAlgorithmNode::work: every frame's first sample index and length (full lists as run-length-free arrays for small n,
    the count and the last three frames for long inputs)"""
    rows = []
    for length, shift, fs in ((400, 160, 16000.0), (200, 80, 8000.0), (160, 160, 16000.0), (100, 160, 16000.0), (7, 3, 100.0)):
        for n in (1, 2, shift - 1, shift, shift + 1, length - 1, length, length + 1, length + shift, length + shift + 1,
                  2 * length - 1, 2 * length, 2 * length + 1, 2 * length + shift, 1000, 4096, 4097, 12345, 48077, 160000):
            if n <= 0:
                continue
            for block in (4096, 1000, 1):
                if block == 1 and n > 5000:
                    continue
                cap = n // shift + 8
                fl, st, first = np.zeros(cap, np.int32), np.zeros(cap, np.float64), np.zeros(cap, np.int64)
                nf = R.ref_time_window_frames(n, block, 3, length, shift, 0, fs, cap, fl.ctypes.data, st.ctypes.data, first.ctypes.data)
                keep = slice(0, nf) if nf <= 12 else slice(nf - 3, nf)
                rows.append(dict(length=length, shift=shift, fs=fs, n=n, block=block, n_frames=int(nf),
                                 first=[int(v) for v in first[keep]], lens=[int(v) for v in fl[keep]],
                                 starts_are_multiples_of_shift=bool(np.all(first[:nf] == np.arange(nf) * shift)),
                                 inner_lens_full=bool(np.all(fl[:max(nf - 3, 0)] == length)),
                                 last_start_time=float(st[nf - 1]).hex()))
    json.dump(rows, open(os.path.join(HERE, "ref_time_framing.json"), "w"), indent=0)


def activation_golden(R):
    """ref_activation.json: Math::mt_vr_exp<f32> (what FastMatrix<f32>::exp() runs inside sigmoid()) on the negated inputs"""
    rng = np.random.Generator(np.random.PCG64(99))
    x = np.concatenate([(rng.standard_normal(4000) * 5).astype(np.float32), np.float32([0, -0.0, 1, -1, 20, -20, 88, -88, 89, -104, 1e-8])])
    nx, y = (-x).astype(np.float32), np.zeros(len(x), np.float32)
    R.ref_mt_vr_exp(len(x), nx.ctypes.data, y.ctypes.data)
    json.dump(dict(x=[float(v).hex() for v in x], exp_neg_x=[float(v).hex() for v in y]), open(os.path.join(HERE, "ref_activation.json"), "w"))


def main():
    R = load_ref()
    if sys.argv[1:] == ["activation"]:
        activation_golden(R)
        return
    if sys.argv[1:] == ["bark"]:
        bark_golden(R)
        return
    if sys.argv[1:] == ["time-framing"]:
        time_framing_golden(R)
        return
    if R is None:
        raise SystemExit("oracle/_ref/libref.so not available (needs /root/reference)")
    rng = np.random.Generator(np.random.PCG64(2024))
    # ---- reference FFT vectors
    fft = {}
    for n in (8, 64, 256, 512, 1024):
        x = (rng.standard_normal(n) * 3000).astype(np.float32)
        y = x.copy()
        R.ref_fft_real(y, n)
        z = x.copy()
        R.ref_fft_complex(z, n)
        fft["in_%d" % n], fft["real_%d" % n], fft["cplx_%d" % n] = x, y, z
    # the survey's probe: sinf(0.1 i), i < 400, zero padded to 512
    x = np.zeros(512, np.float32)
    x[:400] = np.sin((np.float32(0.1) * np.arange(400, dtype=np.float32)).astype(np.float32)).astype(np.float32)
    y = x.copy()
    R.ref_fft_real(y, 512)
    fft["in_sin"], fft["real_sin"] = x, y
    np.savez_compressed(os.path.join(HERE, "ref_fft.npz"), **fft)
    # ---- reference framing (Signal::WindowBuffer driven like SlidingAlgorithmNode)
    framing = []
    for length, shift, fs in ((400, 160, 16000.0), (200, 80, 8000.0), (160, 160, 16000.0), (100, 160, 16000.0)):
        for n in (1, 2, shift - 1, shift, length - 1, length, length + 1, length + shift, length + shift + 1,
                  2 * length, 2 * length + 1, 1000, 4096, 4097, 12345, 48077, 160000):
            if n <= 0:
                continue
            pcm = np.zeros(n, np.float32)
            fl = np.zeros(4000, np.int32)
            st = np.zeros(4000, np.float64)
            for block in (4096, 1000):
                nf = R.ref_window_frames(pcm, n, block, length, shift, fs, 4000, fl.ctypes.data, st.ctypes.data, None)
                framing.append(dict(length=length, shift=shift, fs=fs, n=n, block=block, n_frames=int(nf),
                                    last_len=int(fl[nf - 1]), last_start=float(st[nf - 1]).hex(),
                                    lens_ok=bool(np.all(fl[:nf - 1] == length))))
    json.dump(framing, open(os.path.join(HERE, "ref_framing.json"), "w"), indent=0)
    # ---- reference mel warping / GMM normalisation terms (f64 as hex strings)
    mel = []
    for f in [0.0, 31.25, 62.5, 440.0, 1234.5, 3999.99, 7968.75, 8000.0]:
        mel.append(dict(f=float(f).hex(), mel=float(R.ref_mel(f)).hex(), dmel=float(R.ref_mel_derivative(f)).hex(),
                        inv=float(R.ref_mel_inverse(R.ref_mel(f))).hex()))
    bins = []
    for b in [0, 1, 7, 100, 255, 256]:
        bins.append(dict(bin=b, sr=0.032, warped=float(R.ref_warped_bin(b, 0.032)).hex(),
                         dwarped=float(R.ref_warped_bin_derivative(b, 0.032)).hex(),
                         back=float(R.ref_warped_bin_inverse(R.ref_warped_bin(b, 0.032), 0.032)).hex()))
    var = rng.uniform(0.3, 3.0, 40).astype(np.float32)
    norm = dict(var=[float(v) for v in var], log_norm=float(R.ref_gauss_log_norm_factor(var, 40)).hex(),
                isr=[float(R.ref_inverse_square_root(float(v))) for v in var])
    json.dump(dict(mel=mel, bins=bins, norm=norm), open(os.path.join(HERE, "ref_functions.json"), "w"), indent=0)
    # ---- reference Math::Matrix<f32> * Math::Vector<f32> on the oracle's DCT tables (what Signal::CosineTransform::apply runs)
    #      and |re + i im| of alternating complex vectors through Math::transformAlternatingComplex / pointerAbs
    lin = {}
    for tag, kw in (("16x20", dict(n_ceps=16)), ("40x40", dict(n_ceps=40, filter_width=138.0))):
        m = OracleMfcc(**kw)
        table = np.ascontiguousarray(m.dct, dtype=np.float32)
        x = (rng.standard_normal((6, table.shape[1])) * 2 - 1).astype(np.float32)
        x[5] = 1.0
        y = np.zeros((6, table.shape[0]), np.float32)
        for i in range(6):
            R.ref_matrix_vector(table.ctypes.data, table.shape[0], table.shape[1], x[i].ctypes.data, y[i].ctypes.data)
        lin["dct_table_" + tag], lin["dct_in_" + tag], lin["dct_out_" + tag] = table, x, y
    spec = (rng.standard_normal(514) * np.exp(rng.uniform(-12, 3, 514))).astype(np.float32)
    spec[10:14] = [0.0, 0.0, 3.0, -4.0]
    spec[20:22] = [1e-30, 1e-30]          # squares underflow in f32: hypot does not
    spec[22:24] = [3e20, 4e20]            # squares overflow in f32: hypot does not
    amp = np.zeros(257, np.float32)
    R.ref_complex_amplitude(spec.ctypes.data, 514, amp.ctypes.data)
    lin["spectrum"], lin["amplitude"] = spec, amp
    np.savez_compressed(os.path.join(HERE, "ref_linear.npz"), **lin)
    # ---- SURVEY.md C.1 known answers (reference outputs recorded by the survey)
    c1 = dict(frames_160000=999, last_frame_len=320, w0=0.08, w1=0.08005703,
              filters_268=dict(n=20, mel_max=2840.023047, out0=4.27929, out1=4.360105, out19=4.327153),
              filters_138=dict(n=40, out0=1.994625, out1=2.381036, out39=2.215316),
              fft_sin=dict(x0_re=0.00101768, x1_re=0.000768944, x1_im=-0.000380571),
              dct_20_16_ones=dict(out0=20.0, out1=-1.78814e-07), gmm_score=5.59973)
    json.dump(c1, open(os.path.join(HERE, "survey_c1.json"), "w"), indent=1)
    # ---- oracle outputs on the BASELINE config-1 style inputs
    pcm = synth.waveform(16000, seed=1)
    gold = dict(pcm_s16=pcm.astype(np.int16))
    for tag, kw in (("mfcc16", dict(n_ceps=16)), ("mfcc40", dict(n_ceps=40, filter_width=138.0))):
        m = OracleMfcc(**kw)
        gold[tag] = m.run(pcm)
        st = m.stages(pcm, 3)
        for k, v in st.items():
            gold[tag + "_f3_" + k] = v
    np.savez_compressed(os.path.join(HERE, "orc_mfcc.npz"), **gold)
    model = synth.gmm_cart(32, 1, 8, 40, seed=2, pooled=False)
    x = gold["mfcc40"][:24]
    g = OracleGmm(model)
    sc, best = g.score(x, mode=0)
    ssum, _ = g.score(x, mode=1)
    np.savez_compressed(os.path.join(HERE, "orc_gmm.npz"), feats=x, max_scores=sc, max_best=best, sum_scores=ssum,
                        **{"model_" + k: np.asarray(v) for k, v in model.items()})
    Ws, bs, acts, logp = synth.ffnn([40, 48, 48, 30], seed=7)
    np.savez_compressed(os.path.join(HERE, "orc_ffnn.npz"), feats=x, log_prior=logp,
                        scores64=oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, acc64=1),
                        scores_fma=oracle_ffnn_score(Ws, bs, acts, x, log_prior=logp, acc64=2),
                        **{"W%d" % i: w for i, w in enumerate(Ws)}, **{"b%d" % i: b for i, b in enumerate(bs)})
    bark_golden(R)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
