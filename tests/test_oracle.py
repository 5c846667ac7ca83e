"""CPU: pins the oracle (oracle/liboracle.so) against (a) outputs of the reference itself -- live through
oracle/_ref/libref.so when /root/reference is mounted, and through the committed tests/golden/ref_* vectors
everywhere -- (b) the known answers of SURVEY.md C.1, (c) the reference's unit-test vectors."""
import json
import os

import numpy as np
import pytest

from oracle import MfccCfg, Oracle, OracleGmm, OracleMfcc, load_ref, oracle_ffnn_score
from tests import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
HAVE_REF = os.path.isdir("/root/reference/src")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ----------------------------------------------------------------------------- reference outputs (golden)

def test_fft_bit_exact_against_reference_vectors():
    L = Oracle()
    z = np.load(os.path.join(GOLD, "ref_fft.npz"))
    for n in (8, 64, 256, 512, 1024):
        a = z["in_%d" % n].copy()
        L.orc_fft_real(a, n)
        assert np.array_equal(bits(a), bits(z["real_%d" % n]))
        a = z["in_%d" % n].copy()
        L.orc_fft_complex(a, n)
        assert np.array_equal(bits(a), bits(z["cplx_%d" % n]))
    a = z["in_sin"].copy()
    L.orc_fft_real(a, 512)
    assert np.array_equal(bits(a), bits(z["real_sin"]))


def test_framing_against_reference_window_buffer():
    L = Oracle()
    rows = json.load(open(os.path.join(GOLD, "ref_framing.json")))
    assert len(rows) > 100
    for r in rows:
        cfg = MfccCfg.default()
        cfg.sample_rate = r["fs"]
        cfg.win_len_s = r["length"] / r["fs"]
        cfg.win_shift_s = r["shift"] / r["fs"]
        cfg.fft_max_input_s = max(r["length"], 8) / r["fs"]
        m = OracleMfcc(cfg)
        assert (m.frame_len, m.frame_shift) == (r["length"], r["shift"])
        T = m.n_frames(r["n"])
        assert T == r["n_frames"], r
        last = r["n"] - (T - 1) * r["shift"]
        assert min(r["length"], last) == r["last_len"], r
        assert r["lens_ok"]


def test_temporal_integration_framing_against_reference_time_window_buffer():
    """oracle/orc_gammatone.c: orc_time_window_frames against Signal::TimeWindowBuffer<Flow::Vector<f32>> compiled unmodified
    (tests/golden/ref_time_framing.json, made by make_golden.py time-framing): frame count, every start, every length"""
    from oracle.binding import oracle_time_window_frames
    rows = json.load(open(os.path.join(GOLD, "ref_time_framing.json")))
    assert len(rows) > 200
    for r in rows:
        starts, lens = oracle_time_window_frames(r["n"], r["length"], r["shift"])
        assert len(starts) == r["n_frames"], r
        assert r["starts_are_multiples_of_shift"] and np.array_equal(starts, np.arange(len(starts)) * r["shift"])
        k = len(r["first"])
        assert list(starts[-k:]) == r["first"] and list(lens[-k:]) == r["lens"], r
        assert r["inner_lens_full"] and np.all(lens[:max(len(lens) - 3, 0)] == r["length"])
        assert float.fromhex(r["last_start_time"]) == pytest.approx(starts[-1] / r["fs"], rel=1e-9, abs=1e-9)   # the class accumulates shift / rate


def test_sigmoid_against_reference_exp_template():
    """orc_activation (sigmoid) = (f32)(1.0 / (1.0 + e)), e = Math::mt_vr_exp<f32>(-x) compiled unmodified (ref_activation.json): the
    unqualified exp on a float inside that template is ::exp(double) narrowed, not expf"""
    from oracle.binding import oracle_activation
    g = json.load(open(os.path.join(GOLD, "ref_activation.json")))
    x = np.array([float.fromhex(v) for v in g["x"]], np.float32)
    e = np.array([float.fromhex(v) for v in g["exp_neg_x"]], np.float32)
    with np.errstate(over="ignore"):
        want = (1.0 / (1.0 + e.astype(np.float64))).astype(np.float32)
    assert np.array_equal(bits(oracle_activation(x, 2)), bits(want))
    assert np.array_equal(oracle_activation(x, 1), np.maximum(x, 0))


def test_mel_functions_against_reference_functors():
    L = Oracle()
    g = json.load(open(os.path.join(GOLD, "ref_functions.json")))
    for r in g["mel"]:
        f = float.fromhex(r["f"])
        assert L.orc_mel(f) == float.fromhex(r["mel"])
        assert L.orc_mel_derivative(f) == float.fromhex(r["dmel"])
        assert L.orc_mel_inverse(L.orc_mel(f)) == float.fromhex(r["inv"])
    for r in g["bins"]:   # nest(mel, scale(1/0.032)) as the filter builder composes it
        d2c = 1 / r["sr"]
        assert L.orc_mel(d2c * r["bin"]) == float.fromhex(r["warped"])
        assert L.orc_mel_derivative(d2c * r["bin"]) == float.fromhex(r["dwarped"])
        assert (1 / d2c) * L.orc_mel_inverse(L.orc_mel(d2c * r["bin"])) == float.fromhex(r["back"])


def test_bark_and_equal_loudness_against_reference_functors():
    """bark warping (value, derivative, inverse; alone and nested with disc-to-cont like FilterBuilder::create) and both
    equal-loudness curves against the reference's analytic-function classes compiled unmodified (ref_bark.json, f64 bit-exact);
    plp.flow's composed f(index) equals the oracle's equal-loudness table for both documented spacings"""
    from oracle.binding import MfccCfg
    L = Oracle()
    g = json.load(open(os.path.join(GOLD, "ref_bark.json")))
    for r in g["bark"]:
        f = float.fromhex(r["f"])
        assert L.orc_bark(f) == float.fromhex(r["bark"])
        assert L.orc_bark_derivative(f) == float.fromhex(r["dbark"])
        assert L.orc_bark_inverse(L.orc_bark(f)) == float.fromhex(r["inv"])
        assert L.orc_equal_loudness(f) == float.fromhex(r["eql"])
        assert L.orc_equal_loudness_4khz(f) == float.fromhex(r["eql4k"])
    for r in g["bins"]:
        d2c = 1 / r["sr"]
        assert L.orc_bark(d2c * r["bin"]) == float.fromhex(r["warped"])
        assert L.orc_bark_derivative(d2c * r["bin"]) == float.fromhex(r["dwarped"])
        assert (1 / d2c) * L.orc_bark_inverse(L.orc_bark(d2c * r["bin"])) == float.fromhex(r["back"])
    for r, cfg in zip(g["plp_f"], (MfccCfg.plp(), MfccCfg.plp(n_ceps=11, n_autocorrelation=11, spacing=0.973442, sample_rate=8000.0))):
        m = OracleMfcc(cfg)
        want = np.array([float.fromhex(v) for v in r["values"]])
        assert m.n_transform_inputs == len(want)
        assert np.array_equal(m.equal_loudness, want)        # also pins the choice of the 4 kHz curve for the 8 kHz front-end


def test_plp_filter_bank_known_answers():
    """the reference's own plp.flow documents its filter bank (Tools/FeatureExtraction/share/plp.flow:24-25):
    '8000 Hz -> 19.708905 Bark; #filters 20 -> spacing = 0.93853; 4000 Hz -> 15.575071 Bark; #filters 15 -> spacing = 0.973442'"""
    from oracle.binding import MfccCfg
    m = OracleMfcc(MfccCfg.plp())
    assert m.n_filters == 20 and abs(m.mel_max - 19.708905) < 1e-6 and (m.frame_len, m.fft_len) == (320, 512)
    m = OracleMfcc(MfccCfg.plp(n_ceps=11, n_autocorrelation=11, spacing=0.973442, sample_rate=8000.0))
    assert m.n_filters == 15 and abs(m.mel_max - 15.575071) < 1e-6 and (m.frame_len, m.fft_len) == (160, 256)
    # trapeze geometry: a filter's weights are the f32 shape value times the f64 bark derivative; inside the flat top the shape
    # is exactly 1, so the weight is the derivative itself
    L = Oracle()
    m = OracleMfcc(MfccCfg.plp())
    s, e, o, w = m.filters
    d2c = 1 / 0.032
    for i in (3, 10, 19):
        centre = 0.93853 * (i + 1)
        flat = [b for b in range(s[i], e[i]) if abs(L.orc_bark(d2c * b) - centre) <= 0.5 - 1e-9]
        assert flat, i
        for b in flat:
            assert w[o[i] + b - s[i]] == np.float32(L.orc_bark_derivative(d2c * b))
        # left flank rises by a factor 10 per bark, right flank falls by 10^2.5 per bark
        b = s[i] + 1
        if L.orc_bark(d2c * b) - centre < -0.5:
            rel = L.orc_bark(d2c * b) - centre
            left = -(0.5 / (1.3 - (-2.5))) * 3.8
            assert w[o[i] + 1] == np.float32(float(np.float32(10.0 ** (rel - left))) * L.orc_bark_derivative(d2c * b))


def test_plp_oracle_outputs_unchanged():
    """orc_plp.npz: the oracle's plp.flow outputs recorded after pinning -- drift detector for the GPU box"""
    from oracle.binding import MfccCfg
    g = np.load(os.path.join(GOLD, "orc_plp.npz"))
    pcm = synth.waveform(16000, seed=1)
    for tag, cfg in (("plp16k", MfccCfg.plp(n_ceps=13, n_autocorrelation=13)),
                     ("plp8k", MfccCfg.plp(n_ceps=11, n_autocorrelation=11, spacing=0.973442, sample_rate=8000.0))):
        m = OracleMfcc(cfg)
        assert np.array_equal(bits(m.run(pcm)), bits(g[tag]))
        assert np.array_equal(m.filters[3].view(np.uint32), g[tag + "_fweights"].view(np.uint32))


def test_gmm_normalisation_terms_against_reference_templates():
    g = json.load(open(os.path.join(GOLD, "ref_functions.json")))["norm"]
    var = np.array(g["var"], np.float32)
    model = synth.gmm_cart(2, 1, 1, 40, seed=1)
    model["variances"] = var.reshape(1, 40).copy()
    m2lw, isr, ln = OracleGmm(model).tables()
    assert ln[0] == np.float32(float.fromhex(g["log_norm"]))
    assert np.array_equal(bits(isr[0]), bits(np.array(g["isr"], np.float32)))


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not mounted (GPU box)")
def test_live_reference_build_agrees():
    """Same checks against the freshly built reference TUs, on fresh random inputs."""
    L, R = Oracle(), load_ref()
    assert R is not None
    rng = np.random.Generator(np.random.PCG64(77))
    for n in (16, 128, 512, 2048):
        x = (rng.standard_normal(n) * 1e4).astype(np.float32)
        a, b = x.copy(), x.copy()
        L.orc_fft_real(a, n)
        R.ref_fft_real(b, n)
        assert np.array_equal(bits(a), bits(b))
    m = OracleMfcc(n_ceps=16)
    for n in rng.integers(1, 60000, 40):
        n = int(n)
        fl = np.zeros(1000, np.int32)
        nf = R.ref_window_frames(np.zeros(n, np.float32), n, 4096, 400, 160, 16000.0, 1000, fl.ctypes.data, None, None)
        assert nf == m.n_frames(n)
        assert fl[nf - 1] == min(400, n - (nf - 1) * 160)
    from oracle.binding import oracle_activation
    xs = (rng.standard_normal(3000) * 6).astype(np.float32)
    nx, e = (-xs).astype(np.float32), np.zeros(3000, np.float32)
    R.ref_mt_vr_exp(3000, nx.ctypes.data, e.ctypes.data)
    assert np.array_equal(bits(oracle_activation(xs, 2)), bits((1.0 / (1.0 + e.astype(np.float64))).astype(np.float32)))
    from oracle.binding import oracle_time_window_frames
    for _ in range(60):   # Signal::TimeWindowBuffer (temporal integration) on fresh lengths / shifts / block sizes
        length, shift = int(rng.integers(1, 500)), int(rng.integers(1, 300))
        n, block = int(rng.integers(1, 20000)), int(rng.integers(1, 5000))
        cap = n // shift + 8
        fl, first = np.zeros(cap, np.int32), np.zeros(cap, np.int64)
        nf = R.ref_time_window_frames(n, block, 2, length, shift, 0, 16000.0, cap, fl.ctypes.data, None, first.ctypes.data)
        starts, lens = oracle_time_window_frames(n, length, shift)
        assert nf == len(starts) and np.array_equal(first[:nf], starts) and np.array_equal(fl[:nf], lens), (n, length, shift, block)
    for f in rng.uniform(0, 8000, 50):
        assert L.orc_mel(f) == R.ref_mel(f) and L.orc_mel_derivative(f) == R.ref_mel_derivative(f)
        assert L.orc_mel_inverse(L.orc_mel(f)) == R.ref_mel_inverse(R.ref_mel(f))
    for _ in range(2000):   # Core::isAlmostEqual / isSignificantlyGreater incl. one-ulp neighbours, zero, the tolerance plp.flow uses
        a = float(rng.choice([0.0, 1.0, 9.0, 4000.0, rng.uniform(-10, 10), rng.uniform(0, 1e6)]))
        b = float(rng.choice([a, np.nextafter(a, np.inf), np.nextafter(a, -np.inf), a * (1 + 3e-16), a + rng.uniform(-1e-3, 1e-3), a + 2.0]))
        tol = float(rng.choice([1.0, 2.0, 1e12]))
        assert L.orc_core_is_almost_equal(a, b, tol) == R.ref_is_almost_equal(a, b, tol), (a, b, tol)
        assert L.orc_core_is_significantly_greater(a, b, tol) == R.ref_is_significantly_greater(a, b, tol), (a, b, tol)
    for f in rng.uniform(0, 8000, 50):
        assert L.orc_bark(f) == R.ref_bark(f) and L.orc_bark_derivative(f) == R.ref_bark_derivative(f)
        assert L.orc_bark_inverse(L.orc_bark(f)) == R.ref_bark_inverse(R.ref_bark(f))
        assert L.orc_equal_loudness(f) == R.ref_equal_loudness(f, 0) and L.orc_equal_loudness_4khz(f) == R.ref_equal_loudness(f, 1)
    v = rng.uniform(0.1, 5, 39).astype(np.float32)
    model = synth.gmm_cart(1, 1, 1, 39, seed=3)
    model["variances"] = v.reshape(1, -1).copy()
    _, isr, ln = OracleGmm(model).tables()
    assert ln[0] == np.float32(R.ref_gauss_log_norm_factor(v, 39))
    assert all(isr[0][i] == np.float32(R.ref_inverse_square_root(float(v[i]))) for i in range(39))


def _oracle_dct_apply(table, x):
    """the oracle's DCT stage restated: out[k] = sum_n T[k][n] x[n], f32 products accumulated left to right from 0
    (oracle/orc_mfcc.c: orc_mfcc_frame, following Math/Vector.hh:94-101 as used by Signal/CosineTransform.cc:123-158)"""
    out = np.zeros(table.shape[0], np.float32)
    for k in range(table.shape[0]):
        acc = np.float32(0)
        for n in range(table.shape[1]):
            acc = np.float32(acc + np.float32(table[k, n] * x[n]))
        out[k] = acc
    return out


def test_dct_apply_and_amplitude_against_reference_classes():
    """Math::Matrix<f32> * Math::Vector<f32> (the reference's own classes, compiled unmodified) on the oracle's cosine tables, and
    Math::transformAlternatingComplex + pointerAbs on spectra with entries whose squares under- / overflow in f32:
    golden outputs of the reference (tests/golden/ref_linear.npz) -- the oracle's DCT stage and amplitude stage are bit-identical,
    through the oracle's own frame pipeline as well (the cepstra of a frame = reference product of its log filter-bank row)"""
    L = Oracle()
    g = np.load(os.path.join(GOLD, "ref_linear.npz"))
    for tag, kw in (("16x20", dict(n_ceps=16)), ("40x40", dict(n_ceps=40, filter_width=138.0))):
        m = OracleMfcc(**kw)
        table = np.ascontiguousarray(m.dct, dtype=np.float32)
        assert np.array_equal(bits(table), bits(g["dct_table_" + tag]))
        for x, y in zip(g["dct_in_" + tag], g["dct_out_" + tag]):
            assert np.array_equal(bits(_oracle_dct_apply(table, x)), bits(y))
        pcm = synth.waveform(4000, seed=11)
        st = m.stages(pcm, 5)
        assert np.array_equal(bits(_oracle_dct_apply(table, st["logmel"])), bits(st["ceps"]))
        assert np.array_equal(bits(np.hypot(st["spectrum"][0::2], st["spectrum"][1::2]).astype(np.float32)), bits(st["amplitude"]))
    spec, amp = g["spectrum"], g["amplitude"]
    got = np.array([np.float32(np.hypot(np.float32(spec[2 * k]), np.float32(spec[2 * k + 1]))) for k in range(257)], np.float32)
    assert np.array_equal(bits(got), bits(amp))        # hypotf == std::abs(std::complex<float>) incl. the under- / overflow cases
    R = load_ref()
    if R is not None and hasattr(R, "ref_matrix_vector"):   # live, fresh inputs
        rng = np.random.Generator(np.random.PCG64(123))
        m = OracleMfcc(n_ceps=40, filter_width=138.0)
        table = np.ascontiguousarray(m.dct, dtype=np.float32)
        for _ in range(20):
            x = (rng.standard_normal(table.shape[1]) * 5).astype(np.float32)
            y = np.zeros(table.shape[0], np.float32)
            R.ref_matrix_vector(table.ctypes.data, table.shape[0], table.shape[1], x.ctypes.data, y.ctypes.data)
            assert np.array_equal(bits(_oracle_dct_apply(table, x)), bits(y))
        s2 = (rng.standard_normal(514) * np.exp(rng.uniform(-20, 20, 514))).astype(np.float32)
        a2 = np.zeros(257, np.float32)
        R.ref_complex_amplitude(s2.ctypes.data, 514, a2.ctypes.data)
        assert np.array_equal(bits(np.hypot(s2[0::2], s2[1::2]).astype(np.float32)), bits(a2))


# ----------------------------------------------------------------------------- SURVEY.md C.1 known answers

def _apply_filters(m, amp):
    s, e, o, w = m.filters
    out = []
    for f in range(m.n_filters):
        acc = np.float32(0)
        for b in range(s[f], e[f]):
            acc = np.float32(acc + np.float32(amp[b] * w[o[f] + b - s[f]]))
        out.append(acc)
    return out


def test_survey_known_answers():
    c1 = json.load(open(os.path.join(GOLD, "survey_c1.json")))
    m = OracleMfcc(n_ceps=16)
    assert m.n_frames(160000) == c1["frames_160000"]
    assert 160000 - 998 * 160 == c1["last_frame_len"]
    w = m.window
    assert w[0] == np.float32(c1["w0"]) and abs(w[1] - c1["w1"]) < 5e-9
    ones = np.ones(257, np.float32)
    f = c1["filters_268"]
    out = _apply_filters(m, ones)
    assert m.n_filters == f["n"] and abs(m.mel_max - f["mel_max"]) < 1e-6
    assert abs(out[0] - f["out0"]) < 5e-6 and abs(out[1] - f["out1"]) < 5e-6 and abs(out[19] - f["out19"]) < 5e-6
    m40 = OracleMfcc(n_ceps=40, filter_width=138.0)
    f = c1["filters_138"]
    out = _apply_filters(m40, ones)
    assert m40.n_filters == f["n"]
    assert abs(out[0] - f["out0"]) < 5e-6 and abs(out[1] - f["out1"]) < 5e-6 and abs(out[39] - f["out39"]) < 5e-6
    d = m.dct
    r0 = np.float32(0)
    r1 = np.float32(0)
    for n in range(20):
        r0 = np.float32(r0 + d[0, n])
        r1 = np.float32(r1 + d[1, n])
    assert r0 == np.float32(c1["dct_20_16_ones"]["out0"]) and abs(r1 - c1["dct_20_16_ones"]["out1"]) < 1e-12
    model = dict(dim=4, mix_offsets=np.array([0, 2], np.uint32), dens_index=np.array([0, 1], np.uint32),
                 log_weight=np.log(np.array([0.25, 0.75])), dens_mean=np.array([0, 1], np.uint32),
                 dens_cov=np.array([0, 0], np.uint32), means=np.array([[0] * 4, [1] * 4], np.float32),
                 variances=np.full((1, 4), 2, np.float32))
    sc, best = OracleGmm(model).score(np.full((1, 4), 0.5, np.float32))
    assert abs(sc[0, 0] - c1["gmm_score"]) < 5e-6 and best[0, 0] == 1
    assert abs(OracleGmm(model).score_batch_float(np.full((1, 4), 0.5, np.float32))[0, 0] - c1["gmm_score"]) < 5e-6


# ----------------------------------------------------------------------------- reference unit-test vectors

def test_nn_forward_reference_unit_tests():
    kat = json.load(open(os.path.join(GOLD, "nn_kat.json")))
    for c in kat["cases"]:
        Ws = [np.array(w, np.float32) for w in c["W"]]
        bs = [np.array(b, np.float32) for b in c["bias"]]
        x = np.array(c["input"], np.float32)
        for acc in (0, 1, 2):
            z = -oracle_ffnn_score(Ws, bs, c["hidden_activation"] + [0], x, acc64=acc).astype(np.float64)
            e = np.exp(z - z.max(1, keepdims=True))
            assert np.allclose(e / e.sum(1, keepdims=True), np.array(c["softmax"]), atol=c["tol"])
            if "linear" in c:
                assert np.allclose(z, np.array(c["linear"]), atol=1e-6)
            if "sigmoid" in c:
                assert np.allclose(1 / (1 + np.exp(-z)), np.array(c["sigmoid"]), atol=c["tol"])


# ----------------------------------------------------------------------------- oracle self-consistency / drift

def test_oracle_outputs_have_not_drifted():
    z = np.load(os.path.join(GOLD, "orc_mfcc.npz"))
    pcm = z["pcm_s16"].astype(np.float32)
    assert np.array_equal(pcm, synth.waveform(16000, seed=1))
    for tag, kw in (("mfcc16", dict(n_ceps=16)), ("mfcc40", dict(n_ceps=40, filter_width=138.0))):
        m = OracleMfcc(**kw)
        assert np.array_equal(bits(m.run(pcm)), bits(z[tag]))
        st = m.stages(pcm, 3)
        for k, v in st.items():
            assert np.array_equal(bits(v), bits(z[tag + "_f3_" + k])), (tag, k)
    g = np.load(os.path.join(GOLD, "orc_gmm.npz"))
    model = {k[6:]: g[k] for k in g.files if k.startswith("model_")}
    model["dim"] = int(model["dim"])
    o = OracleGmm(model)
    sc, best = o.score(g["feats"], mode=0)
    assert np.array_equal(bits(sc), bits(g["max_scores"])) and np.array_equal(best, g["max_best"])
    assert np.array_equal(bits(o.score(g["feats"], mode=1)[0]), bits(g["sum_scores"]))


def test_mfcc_against_numpy_restatement():
    """independent float64 numpy computation of the same chain (rfft based): agreement to ~1e-5 shows the oracle
    computes MFCCs, not merely something self-consistent."""
    pcm = synth.waveform(8000, seed=21)
    m = OracleMfcc(n_ceps=40, filter_width=138.0)
    got = m.run(pcm)
    x = pcm.astype(np.float64)
    y = np.concatenate([[0.0], x[1:] - x[:-1]])
    s, e, o, w = m.filters
    win = m.window.astype(np.float64)
    dct = m.dct.astype(np.float64)
    T = m.n_frames(len(pcm))
    ref = np.zeros((T, 40))
    for t in range(T):
        fr = y[t * 160:t * 160 + 400]
        buf = np.zeros(512)
        buf[:len(fr)] = fr * win[:len(fr)]
        amp = np.abs(np.fft.rfft(buf)) / 16000.0
        mel = np.array([np.dot(amp[s[f]:e[f]], w[o[f]:o[f + 1]].astype(np.float64)) for f in range(40)])
        ref[t] = dct @ np.log10(mel)
    assert np.allclose(got, ref, rtol=2e-5, atol=2e-4)


def test_gmm_max_against_numpy_restatement():
    model = synth.gmm_cart(30, 1, 6, 24, seed=8, pooled=False)
    x = np.random.Generator(np.random.PCG64(9)).standard_normal((20, 24)).astype(np.float32)
    sc, best = OracleGmm(model).score(x)
    mu, var = model["means"].astype(np.float64), model["variances"].astype(np.float64)
    for m in range(30):
        k0, k1 = model["mix_offsets"][m], model["mix_offsets"][m + 1]
        d = model["dens_index"][k0:k1]
        ll = (-2 * model["log_weight"][k0:k1])[None, :] + (24 * np.log(2 * np.pi) + np.log(var[d]).sum(1))[None, :] + \
             (((x[:, None, :].astype(np.float64) - mu[d][None]) ** 2) / var[d][None]).sum(2)
        assert np.allclose(sc[:, m], 0.5 * ll.min(1), rtol=1e-5)
        assert np.array_equal(best[:, m], ll.argmin(1))
    ssum, _ = OracleGmm(model).score(x, mode=1)
    for m in range(30):
        k0, k1 = model["mix_offsets"][m], model["mix_offsets"][m + 1]
        d = model["dens_index"][k0:k1]
        ll = 0.5 * ((-2 * model["log_weight"][k0:k1])[None, :] + (24 * np.log(2 * np.pi) + np.log(var[d]).sum(1))[None, :] +
                    (((x[:, None, :].astype(np.float64) - mu[d][None]) ** 2) / var[d][None]).sum(2))
        lse = -np.log(np.exp(-(ll - ll.min(1, keepdims=True))).sum(1)) + ll.min(1)
        assert np.allclose(ssum[:, m], lse, rtol=1e-5, atol=1e-5)


def test_batch_float_scorer_close_to_diagonal_maximum():
    model = synth.gmm_cart(40, 1, 8, 40, seed=18, pooled=True)
    x = np.random.Generator(np.random.PCG64(19)).standard_normal((10, 40)).astype(np.float32)
    o = OracleGmm(model)
    assert np.allclose(o.score_batch_float(x), o.score(x)[0], rtol=2e-6)
    with pytest.raises(ValueError):
        OracleGmm(synth.gmm_cart(4, 1, 2, 8, seed=1, pooled=False)).score_batch_float(np.zeros((1, 8), np.float32))


def test_quantizer_matches_reference_functor():
    """quantize<f32, u8> of the SIMD-diagonal-maximum scorer: oracle == the reference's functor (libref, when built) on the rounding
    boundaries, the clipping limits and the out-of-range inputs whose conversion is undefined in C (x86: INT_MIN)."""
    from oracle.binding import load_ref, oracle_quantize, ref_quantize
    vals = [0.0, 0.4, 0.5, -0.5, 1.5, -1.5, 2.5, 126.49, 126.5, 127.4, 127.5, 200.0, -127.5, -128.4, -128.5, -129.0, 1e10, -1e10,
            float("inf"), float("-inf"), float("nan"), 3e9, -3e9, 2147483520.0, -2147483648.0]
    known = {0.0: 128, 0.5: 129, -0.5: 127, 1.5: 130, 2.5: 131, 126.5: 255, 127.5: 255, -128.5: 0, 1e10: 0, -1e10: 0}
    for v in vals:
        q = oracle_quantize(v)
        if v in known:
            assert q == known[v], (v, q)
        if load_ref() is not None:
            assert q == ref_quantize(v), (v, q, ref_quantize(v))


def test_levinson_matches_reference_golden_and_live():
    """Math::LevinsonLeastSquares: the oracle's restatement against outputs of the reference's own translation unit (golden vectors
    from libref.so, and libref.so itself when present), bit for bit, including the recursions that fail."""
    import json
    import os
    from oracle.binding import load_ref, oracle_levinson, ref_levinson
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_levinson.json")))
    failures = 0
    for c in g["cases"]:
        R = np.frombuffer(bytes.fromhex(c["R"]), "<f4")
        out = oracle_levinson(R)
        assert (out is not None) == c["ok"]
        if out is None:
            failures += 1
            continue
        assert np.float32(out[0]).tobytes().hex() == c["gain"] and out[1].astype("<f4").tobytes().hex() == c["a"]
    assert failures >= 1
    if load_ref() is not None:
        rng = np.random.Generator(np.random.PCG64(2))
        for _ in range(200):
            n = int(rng.integers(2, 30))
            x = rng.standard_normal(300)
            R = np.array([np.dot(x[:300 - k], x[k:]) for k in range(n)], np.float32)
            a, b = oracle_levinson(R), ref_levinson(R)
            assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


def test_lpc_cepstrum_recursion_against_independent_definition():
    """Signal::autoregressionToCepstrum (parity unpinned: its translation unit needs Flow): the recursion must give the cepstrum of
    the all-pole model gain / A(z), i.e. the inverse DFT of log |H|^2 / 2 ... checked here through the power series of
    log(1 / A(z)) computed independently with numpy polynomial arithmetic in f64."""
    from oracle.binding import oracle_ar_to_cepstrum
    rng = np.random.Generator(np.random.PCG64(4))
    for _ in range(20):
        order = int(rng.integers(1, 16))
        # a stable A(z) = prod (1 - r_i z^-1): coefficients a_1..a_N of 1 + a_1 z^-1 + ...
        roots = rng.uniform(-0.8, 0.8, order)
        A = np.poly(roots)                                # [1, a1, ..., aN]
        a = A[1:].astype(np.float32)
        gain = np.float32(rng.uniform(0.1, 5))
        nc = order + 1
        c = oracle_ar_to_cepstrum(gain, a, nc)
        # log(1/A(z)) = sum_n (sum_i r_i^n / n) z^-n ; the reference's convention for c[0] is 2 log(gain)
        want = [2 * np.log(float(gain))] + [float(np.sum(roots ** n) / n) for n in range(1, nc)]
        assert np.allclose(c, want, rtol=2e-4, atol=2e-5), (c, want)


def test_mfplp_oracle_chain_properties():
    """MF-PLP restatement end to end on a synthetic signal: finite output of the configured size, c0 = 2 log(gain) moves by
    2 * 0.33 * 2 * log(s) when the signal is scaled by s (power spectrum ^ 0.33 -> autocorrelation scale s^0.66 -> gain^2),
    higher cepstra are scale invariant; digital silence makes the recursion fail (NaN, where the reference reports an error)."""
    from oracle import OracleMfcc
    from oracle.binding import MfccCfg
    from tests import synth
    pcm = synth.waveform(16000, seed=9)
    m = OracleMfcc(MfccCfg.mfplp(n_ceps=13, n_autocorrelation=15))
    a = m.run(pcm)
    b = m.run(pcm * np.float32(4.0))
    assert a.shape == (99, 13) and np.all(np.isfinite(a))
    assert np.allclose(b[:, 0] - a[:, 0], 2 * 0.33 * np.log(4.0), atol=2e-3)
    assert np.allclose(b[:, 1:], a[:, 1:], atol=5e-4)
    z = m.run(np.zeros(1600, np.float32))
    assert np.all(np.isnan(z))
