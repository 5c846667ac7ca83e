"""CPU: the C-ABI library loads, exports every symbol include/amx.h declares, and its HOST logic (geometry,
table construction, model preparation, .pms IO, error behaviour) agrees with the oracle.  No kernel runs here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import MfccCfg as OrcCfg, OracleGmm, OracleMfcc
from rasr_amd import _lib
from tests import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "amx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(amx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    L = _lib.lib()
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), "librasr_amd.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, set(names) ^ set(_lib.SIGNATURES)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.lib()
    h = C.c_void_p()
    assert L.amx_init(0, C.byref(h)) == _lib.AMX_ERR_DEVICE
    assert b"no CPU fallback" in L.amx_last_error()
    import rasr_amd
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.Context(0)


# plp.flow's parameter values (amx_plp_default_cfg)
PLP = dict(front_end=2, win_len_s=0.02, fft_max_input_s=0.02, preemph_alpha=0.0, mel_filter_width=3.8, mel_spacing=0.93853,
           filter_type=1, boundary=1, warping=1, dct_normalize=1, n_autocorrelation=13, n_ceps=13)


def host_mfcc(**kw):
    L = _lib.lib()
    cfg = _lib.MfccCfg()
    L.amx_mfcc_default_cfg(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    h = C.c_void_p()
    st = L.amx_mfcc_create(None, C.byref(cfg), C.byref(h))
    return L, h, st


@pytest.mark.parametrize("kw", [dict(), dict(n_ceps=40, mel_filter_width=138.0), dict(sample_rate=8000.0),
                                dict(sample_rate=44100.0, n_ceps=13), dict(sample_rate=11025.0, mel_filter_width=150.0),
                                dict(mel_spacing=100.0), dict(warp_differential_unit=0), dict(win_len_s=0.02, win_shift_s=0.0125),
                                dict(front_end=1, n_autocorrelation=13, n_ceps=13, dct_normalize=1),
                                dict(front_end=1, n_autocorrelation=12, n_ceps=9, dct_normalize=1, sample_rate=8000.0),
                                PLP, dict(PLP, sample_rate=8000.0, mel_spacing=0.973442, n_autocorrelation=11, n_ceps=11),
                                dict(PLP, mel_spacing=0.0, n_autocorrelation=9, n_ceps=8), dict(PLP, boundary=2, sample_rate=11025.0),
                                dict(filter_type=1, warping=1, boundary=1, mel_filter_width=3.8, mel_spacing=0.9),
                                dict(warping=1, mel_filter_width=2.0), dict(boundary=2, mel_filter_width=300.0),
                                dict(filter_type=1, mel_filter_width=500.0)])
def test_host_tables_bit_identical_to_oracle(kw):
    L, h, st = host_mfcc(**kw)
    assert st == 0, L.amx_last_error()
    info = _lib.MfccInfo()
    L.amx_mfcc_describe(h, C.byref(info))
    o = OrcCfg.default()
    for k, v in kw.items():
        setattr(o, k, v)
    m = OracleMfcc(o)
    assert (info.frame_len, info.frame_shift, info.fft_len, info.n_bins, info.n_filters, info.n_ceps) == \
           (m.frame_len, m.frame_shift, m.fft_len, m.n_bins, m.n_filters, m.n_ceps)
    assert info.mel_max == m.mel_max
    fo = np.zeros(info.n_filters + 1, np.int32)
    L.amx_mfcc_tables(h, None, None, None, fo.ctypes.data, None, None)
    win, fs, fe = np.zeros(info.frame_len, np.float32), np.zeros(info.n_filters, np.int32), np.zeros(info.n_filters, np.int32)
    assert info.n_transform == (kw["n_autocorrelation"] if kw.get("front_end") else info.n_ceps)
    assert info.n_transform_inputs == m.n_transform_inputs
    fw, dct = np.zeros(fo[-1], np.float32), np.zeros((info.n_transform, info.n_transform_inputs), np.float32)
    L.amx_mfcc_tables(h, win.ctypes.data, fs.ctypes.data, fe.ctypes.data, fo.ctypes.data, fw.ctypes.data, dct.ctypes.data)
    s, e, off, w = m.filters
    assert np.array_equal(win.view(np.uint32), m.window.view(np.uint32))
    assert np.array_equal(fs, s) and np.array_equal(fe, e) and np.array_equal(fo, off)
    assert np.array_equal(fw.view(np.uint32), w.view(np.uint32))
    assert np.array_equal(dct.view(np.uint32), m.dct.view(np.uint32))
    eql = np.zeros(info.n_transform_inputs, np.float64)
    if kw.get("front_end") == 2:
        assert L.amx_mfcc_equal_loudness(h, eql.ctypes.data) == 0 and np.array_equal(eql, m.equal_loudness)
    else:
        assert L.amx_mfcc_equal_loudness(h, eql.ctypes.data) == _lib.AMX_ERR_STATE
    for n in (0, 1, 399, 400, 401, 5000, 160000, 123457):
        assert L.amx_mfcc_n_frames(h, n) == m.n_frames(n)
    # frame start times accumulate like WindowBuffer (bufferStartTime_ += shift/fs)
    t = 0.0
    for k in range(1000):
        if k in (0, 1, 17, 998):
            assert L.amx_mfcc_frame_start_time(h, k) == t
        t += info.frame_shift / o.sample_rate
    # a host-only handle cannot run
    out = np.zeros(16, np.float32)
    assert L.amx_mfcc_run(h, out.ctypes.data, 4, out.ctypes.data) == _lib.AMX_ERR_STATE
    L.amx_mfcc_destroy(h)


def test_mfcc_configuration_errors():
    L, h, st = host_mfcc(sample_rate=0.0)
    assert st == _lib.AMX_ERR_INVALID and b"not positive" in L.amx_last_error()
    # window longer than the FFT: the reference node raises "Input data size ... is larger then maximal input size"
    L, h, st = host_mfcc(win_len_s=0.05)
    assert st == _lib.AMX_ERR_INVALID and b"larger then maximal input size" in L.amx_last_error()
    L, h, st = host_mfcc(n_ceps=0)
    assert st == _lib.AMX_ERR_INVALID
    # MF-PLP: the cosine transform cannot produce more outputs than the filter bank has, the cepstrum at most order + 1
    L, h, st = host_mfcc(front_end=1, n_autocorrelation=20, n_ceps=9, sample_rate=8000.0)
    assert st == _lib.AMX_ERR_INVALID and b"nr-autocorrelation-coefficients" in L.amx_last_error()
    L, h, st = host_mfcc(front_end=1, n_autocorrelation=10, n_ceps=11)
    assert st == _lib.AMX_ERR_INVALID and b"Incorrect output size" in L.amx_last_error()
    L, h, st = host_mfcc(front_end=7)
    assert st == _lib.AMX_ERR_INVALID


def test_gmm_prepared_tables_bit_identical_to_oracle():
    import rasr_amd
    L = _lib.lib()
    # gaussian-scale 0.9: the f64 root narrowed to f32 (what the reference's constructor does) differs from a root taken in f32
    for pooled, mws, gsc in ((True, 1.0, 1.0), (False, 0.7, 1.3), (False, 0.3, 0.9)):
        model = synth.gmm_cart(50, 1, 8, 40, seed=2, pooled=pooled)
        keep = []
        st = rasr_amd._gmm_struct(model, mws, gsc, keep)
        h = C.c_void_p()
        assert L.amx_gmm_create(None, C.byref(st), C.byref(h)) == 0, L.amx_last_error()
        nk, nc = int(model["mix_offsets"][-1]), model["variances"].shape[0]
        a, b, c = np.zeros(nk, np.float32), np.zeros((nc, 40), np.float32), np.zeros(nc, np.float32)
        L.amx_gmm_tables(h, a.ctypes.data, b.ctypes.data, c.ctypes.data)
        oa, ob, oc = OracleGmm(model, mws, gsc).tables()
        assert np.array_equal(a.view(np.uint32), oa.view(np.uint32))
        assert np.array_equal(b.view(np.uint32), ob.view(np.uint32))
        assert np.array_equal(c.view(np.uint32), oc.view(np.uint32))
        if gsc == 0.9:   # inverse root of the first variance times (f32)sqrt((f64)0.9), from the definition (CovarianceFeatureScorerElement::scale)
            gs = np.float32(np.sqrt(np.float64(0.9)))
            assert gs != np.sqrt(np.float32(0.9))
            assert ob[0, 0] == np.float32(np.float32(1.0) / np.float32(np.sqrt(np.float64(model["variances"][0, 0])))) * gs
        assert L.amx_gmm_n_mixtures(h) == 50 and L.amx_gmm_dimension(h) == 40
        x = np.zeros((1, 40), np.float32)
        assert L.amx_gmm_score(h, 0, x.ctypes.data, 1, x.ctypes.data, None) == _lib.AMX_ERR_STATE
        L.amx_gmm_destroy(h)


def test_gmm_model_validation():
    import rasr_amd
    L = _lib.lib()
    model = synth.gmm_cart(4, 1, 2, 8, seed=1)
    bad = dict(model)
    bad["dens_index"] = model["dens_index"].copy()
    bad["dens_index"][0] = 999
    keep = []
    h = C.c_void_p()
    assert L.amx_gmm_create(None, C.byref(rasr_amd._gmm_struct(bad, 1.0, 1.0, keep)), C.byref(h)) == _lib.AMX_ERR_INVALID
    bad = dict(model)
    bad["variances"] = -model["variances"]
    assert L.amx_gmm_create(None, C.byref(rasr_amd._gmm_struct(bad, 1.0, 1.0, keep)), C.byref(h)) == _lib.AMX_ERR_INVALID
    assert b"variance" in L.amx_last_error()


def test_pms_round_trip(tmp_path):
    import rasr_amd
    model = synth.gmm_cart(12, 1, 5, 7, seed=3, pooled=False)
    p = str(tmp_path / "m.pms")
    rasr_amd.write_pms(model, p)
    head = open(p).read().split("\n")[:3]
    assert head[0] == "#Version: 2.0" and head[1] == "#CovarianceType: DiagonalCovariance"
    assert head[2].split() == ["7", "12", str(len(model["dens_mean"])), str(model["means"].shape[0]), str(model["variances"].shape[0])]
    back = rasr_amd.read_pms(p)
    for k in ("mix_offsets", "dens_index", "dens_mean", "dens_cov", "log_weight", "means", "variances"):
        assert np.array_equal(back[k], model[k]), k


def test_pms_reads_reference_style_files(tmp_path):
    """a file as Mm::MixtureSet::write emits it (6 significant digits, weight column), and a version-1 file with
    linear mixture weights and non-unit covariance weights"""
    import rasr_amd
    p = str(tmp_path / "v2.pms")
    open(p, "w").write("#Version: 2.0\n#CovarianceType: DiagonalCovariance\n2 2 3 3 1\n"
                       "2 0 -0.693147 1 -0.693147\n1 2 0\n0 0\n1 0\n2 0\n2 0.5 -1\n2 1.5 2\n2 0 0\n 2 2 1 0.5 1\n")
    m = rasr_amd.read_pms(p)
    assert m["dim"] == 2 and list(m["mix_offsets"]) == [0, 2, 3] and list(m["dens_index"]) == [0, 1, 2]
    assert np.allclose(m["log_weight"], [-0.693147, -0.693147, 0]) and np.allclose(m["variances"], [[2, 0.5]])
    p1 = str(tmp_path / "v1.pms")
    open(p1, "w").write("#Version: 1.0\n#CovarianceType: DiagonalCovariance\n1 1 2 2 1\n2 0 0.25 1 0.75\n0 0\n1 0\n1 0\n1 1\n 1 2 1.5\n")
    m = rasr_amd.read_pms(p1)
    assert np.allclose(m["log_weight"], np.log([0.25, 0.75])) and np.allclose(m["variances"], [[3.0]])
    bad = str(tmp_path / "bad.pms")
    open(bad, "w").write("#Version: 3.0\n#CovarianceType: DiagonalCovariance\n")
    with pytest.raises(rasr_amd.AmxError) as e:
        rasr_amd.read_pms(bad)
    assert e.value.status == _lib.AMX_ERR_UNSUPPORTED
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.read_pms(str(tmp_path / "missing.pms"))


def test_tuning_strings_are_parsed_strictly_and_nothing_reads_kernel_switches_from_the_environment():
    """amx_*_model.tuning: "key=value,..." chooses kernels for A/B runs and tests; a key the handle does not know or an item that is
    not key=value fails the creation (a typo must not silently give the default kernel).  The shipped library reads no AMX_*
    switch from the environment except AMX_RCCL_LIB (the file name of the RCCL library, include/amx.h)."""
    import subprocess

    import rasr_amd
    model = synth.gmm_cart(4, 1, 2, 8, seed=1)
    rasr_amd.GmmFeatureScorer(None, model, tuning="screen=0, fused=0,tied_prune=1")      # host-only handle: parsed, unused
    rasr_amd.GmmFeatureScorer(None, model, tuning={"chunk": 4096, "screen_kernel": "persist"})
    for bad in ("scren=0", "screen", "screen=0,=1", "tile=3"):
        with pytest.raises(rasr_amd.AmxError, match="tuning"):
            rasr_amd.GmmFeatureScorer(None, model, tuning=bad)
    out = subprocess.run(["strings", _lib.LIB_PATH], capture_output=True, text=True).stdout.split()
    env_names = sorted({w for w in out if w.startswith("AMX_") and w.isupper() and not w.startswith(("AMX_PREC_", "AMX_CONTRACT_"))})   # enum names in error texts
    assert env_names == ["AMX_RCCL_LIB"], env_names
    srcs = os.path.join(ROOT, "rasr_amd", "csrc")
    for f in os.listdir(srcs):
        if f.endswith((".hip", ".cpp", ".hpp")):
            for n, line in enumerate(open(os.path.join(srcs, f), errors="replace"), 1):
                if "getenv" in line:
                    assert "AMX_RCCL_LIB" in line or "AMX_TUNING" in line, "%s:%d reads the environment: %s" % (f, n, line.strip())


def test_product_does_not_reference_the_oracle():
    """the shipped package must never import, link or call anything under oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rasr_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".hh")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower() or f == "__init__.py" and "oracle" not in text, os.path.join(dirpath, f)
    out = os.popen("ldd %s" % _lib.LIB_PATH).read()
    assert "liboracle" not in out and "libref" not in out


def test_nn_parameter_file_round_trip_and_layout(tmp_path):
    """the reference's own layout test (Test/Nn_LinearLayer.cc:43-63): parameter matrix [out x (1+in)], column 0 = bias"""
    import struct

    import rasr_amd
    params = np.arange(12, dtype=np.float32).reshape(3, 4) * 0.5 - 1
    p = str(tmp_path / "net-f32-layer-1.bin")
    rasr_amd.write_nn_matrix("bin:" + p, params)
    raw = open(p, "rb").read()
    assert struct.unpack("<III", raw[:12]) == (3, 4, 3)                     # Matrix::write + vector<Vector> header
    assert struct.unpack("<I", raw[12:16]) == (4,) and len(raw) == 12 + 3 * (4 + 16)
    assert struct.unpack("<4f", raw[16:32]) == tuple(params[0])
    back = rasr_amd.read_nn_matrix(p)
    assert np.array_equal(back, params)
    W, b = rasr_amd.layer_from_parameters(back, has_bias=True)
    assert np.array_equal(b, params[:, 0]) and np.array_equal(W, params[:, 1:])
    W2, b2 = rasr_amd.layer_from_parameters(back, has_bias=False)
    assert np.array_equal(W2, params) and not b2.any()
    open(p, "wb").write(raw[:-3])
    with pytest.raises(rasr_amd.AmxError):
        rasr_amd.read_nn_matrix(p)


def test_prior_from_mixture_set():
    import rasr_amd
    model = synth.gmm_cart(40, 1, 6, 8, seed=4)
    got = rasr_amd.prior_from_mixture_set(model)
    # numpy restatement of Nn/Prior.cc:159-188 with the same f32 / f64 steps
    pri = np.zeros(40, np.float32)
    for m in range(40):
        acc = np.float32(0)
        for k in range(model["mix_offsets"][m], model["mix_offsets"][m + 1]):
            acc = np.float32(np.float64(acc) + np.exp(model["log_weight"][k]))
        pri[m] = acc
    obs = np.float32(pri.astype(np.float64).sum())
    want = np.log((pri / obs).astype(np.float32)).astype(np.float32)
    assert np.allclose(got, want, rtol=0, atol=2e-7)
    assert abs(np.exp(got.astype(np.float64)).sum() - 1) < 1e-5


# ------------------------------------------------------------------ signal-dc-detection (host code, no GPU)

def _dc_signal(rng, n):
    """noise with planted constant runs (some longer, some shorter than min-dc-length), near-constant stretches inside the increment,
    and short bursts between two DC runs"""
    x = (rng.standard_normal(n) * rng.choice([30.0, 3000.0])).astype(np.float32)
    for _ in range(int(rng.integers(0, 8))):
        a = int(rng.integers(0, max(1, n - 1)))
        ln = int(rng.choice([5, 150, 199, 200, 201, 400, 3000, 9000]))
        kind = int(rng.integers(0, 3))
        seg = slice(a, min(n, a + ln))
        if kind == 0:
            x[seg] = np.float32(rng.choice([0.0, -32768.0, 32767.0, 12.5]))
        elif kind == 1:
            x[seg] = np.float32(100.0) + (rng.uniform(-0.4, 0.4, seg.stop - seg.start)).astype(np.float32)     # within the increment
        else:
            x[seg] = 0.0
            b = min(n, a + ln // 2)
            e = min(n, b + int(rng.integers(1, 300)))
            if b < e:
                x[b:e] = (rng.standard_normal(e - b) * 500).astype(np.float32)
    return x


def test_dc_detection_matches_the_restatement_and_its_definition():
    """amx_dc_detection against the oracle's literal transcription of Signal::DcDetection (any input vector size gives the same blocks),
    and against what the node documents: accepted blocks are disjoint and ordered, never longer than max(min segment, maximal output
    size) + a DC hypothesis shorter than min-dc-length, dropped stretches are DC runs or short segments, noise passes untouched"""
    import rasr_amd
    from oracle.binding import oracle_dc_detection
    rng = np.random.Generator(np.random.PCG64(81))
    for case in range(120):
        n = int(rng.choice([0, 1, 2, 199, 200, 201, 321, 5000, 40000]))
        fs = float(rng.choice([8000.0, 16000.0]))
        x = _dc_signal(rng, n) if n else np.zeros(0, np.float32)
        kw = dict(sample_rate=fs, min_dc_length=float(rng.choice([0.0125, 0.005, 0.0])), max_dc_increment=float(rng.choice([0.9, 0.0, 5.0])),
                  min_non_dc_segment_length=float(rng.choice([0.02, 0.026, 0.0])), maximal_output_size=int(rng.choice([4096, 256, 1])))
        got = rasr_amd.dc_detection(x, **kw)
        for block in (4096, 1000, 1, int(rng.integers(1, 5000))):
            assert got == oracle_dc_detection(x, block=block, **kw), (case, n, kw, block)
        pos = 0
        for s, l in got:
            assert s >= pos and l >= 1 and s + l <= n
            pos = s + l
        merged = rasr_amd.dc_detection(x, merge=True, **kw)
        assert sum(l for _, l in merged) == sum(l for _, l in got)
        assert all(a[0] + a[1] < b[0] for a, b in zip(merged, merged[1:]))            # a real gap between two merged ranges
    x = (rng.standard_normal(50000) * 3000).astype(np.float32)
    assert rasr_amd.dc_detection(x, merge=True) == [(0, 50000)]                      # nothing to drop
    x[10000:10400] = 7.0
    assert rasr_amd.dc_detection(x, merge=True) == [(0, 10001), (10400, 39600)]      # the first constant sample still differs from its predecessor
    x[10400:10500] = (rng.standard_normal(100) * 3000).astype(np.float32)
    x[10500:11000] = -3.0                                                            # 100 samples between two DC runs: shorter than 20 ms
    assert rasr_amd.dc_detection(x, merge=True) == [(0, 10001), (11000, 39000)]
    assert rasr_amd.dc_detection(np.zeros(5000, np.float32)) == []                   # digital silence: one sample + a DC run, too short to keep
    assert rasr_amd.dc_detection(np.zeros(5000, np.float32), max_dc_increment=0.0, merge=True) == [(0, 5000)]    # detection disabled


def test_vector_files_accept_the_attribute_forms_an_xml_parser_accepts(tmp_path):
    """amx_nn_vector_read_*: the reference reads Math::Vector files through a real XML parser (Core/VectorParser.hh), so
    `size = "3"`, single quotes, line breaks inside the tag and a commented-out copy of the element in front of it are all the
    same document; a wrong size is still the reference's "Vector dimension mismatch" error"""
    L = _lib.lib()
    heads = ['<vector-f32 size="3">', '<vector-f32 size = "3">', "<vector-f32 size='3'>", '<vector-f32  size =\n "3" >',
             '<!-- <vector-f32 size="9"> 1 </vector-f32> -->\n<vector-f32 size="3">', '<vector-f32>']
    for i, head in enumerate(heads):
        p = str(tmp_path / ("v%d.xml" % i))
        with open(p, "w") as f:
            f.write('<?xml version="1.0" encoding="ISO-8859-1"?>\n' + head + " 1.5 -2 3e1 </vector-f32>\n")
        n, ptr = C.c_int(), C.c_void_p()
        assert L.amx_nn_vector_read_f32(p.encode(), C.byref(n), C.byref(ptr)) == 0, (head, L.amx_last_error())
        v = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n.value,)).copy()
        L.amx_free(ptr)
        assert v.tolist() == [1.5, -2.0, 30.0], head
    p = str(tmp_path / "bad.xml")
    with open(p, "w") as f:
        f.write('<vector-f32 size = "4"> 1 2 3 </vector-f32>\n')
    n, ptr = C.c_int(), C.c_void_p()
    assert L.amx_nn_vector_read_f32(p.encode(), C.byref(n), C.byref(ptr)) == _lib.AMX_ERR_INVALID
    assert b"Vector dimension mismatch: 4 given and 3 read" in L.amx_last_error()


def test_contract_macro_follows_the_including_translation_units_flags(tmp_path):
    """AMX_CONTRACT_OF_THIS_BUILD is decided where include/amx.h is INCLUDED (the adapter, built with RASR's flags): FMA with -mfma /
    -march=haswell, OFF for plain x86-64 and when the adapter says its build does not contract; rasr_amd/host's initContext compiles"""
    import subprocess
    src = tmp_path / "m.cc"
    src.write_text('#include "%s/rasr_amd/host/BatchFeatureScorer.hh"\n#include <cstdio>\nint main() { std::printf("%%d\\n", AMX_CONTRACT_OF_THIS_BUILD); '
                   'return &AmxHost::initContext == nullptr; }\n' % ROOT)
    def value(*flags):
        exe = str(tmp_path / "m")
        subprocess.check_call(["g++", "-std=c++17", "-O0", *flags, "-c", str(src), "-o", exe + ".o"])
        out = subprocess.run(["g++", "-std=c++17", "-E", "-dM", *flags, "-include", os.path.join(ROOT, "include", "amx.h"), "-x", "c++", "/dev/null"],
                             capture_output=True, text=True, check=True).stdout
        line = [l for l in out.splitlines() if l.startswith("#define AMX_CONTRACT_OF_THIS_BUILD")][0]
        return line.split()[2]
    assert value("-march=x86-64") == "AMX_CONTRACT_OFF"
    assert value("-march=x86-64", "-mfma") == "AMX_CONTRACT_FMA"
    assert value("-march=haswell") == "AMX_CONTRACT_FMA"
    assert value("-march=haswell", "-DAMX_ADAPTER_NO_FP_CONTRACT") == "AMX_CONTRACT_OFF"
