"""ctypes binding of oracle/liboracle.so (and oracle/_ref/libref.so when present).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The reference has two arithmetics (orc.h): contract "off" = built for plain x86-64 (-msse3), "fma" = its default -march=native build
# on an FMA host.  One oracle library and one flavour of the compiled reference per mode.
CONTRACTS = ("off", "fma")
_LIBS = {"off": os.path.join(_HERE, "liboracle.so"), "fma": os.path.join(_HERE, "liboracle_fma.so")}
_REFS = {"off": os.path.join(_HERE, "_ref", "libref.so"), "fma": os.path.join(_HERE, "_ref", "libref_native.so")}
_SRCS = ("orc_mfcc.c", "orc_score.c", "orc_backend.c", "orc_gammatone.c")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class MfccCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_double), ("win_len_s", C.c_double), ("win_shift_s", C.c_double),
                ("preemph_alpha", C.c_double), ("fft_max_input_s", C.c_double), ("apply_scale", C.c_int),
                ("mel_filter_width", C.c_double), ("mel_spacing", C.c_double),
                ("warp_differential_unit", C.c_int), ("n_ceps", C.c_int), ("dct_normalize", C.c_int),
                ("front_end", C.c_int), ("n_autocorrelation", C.c_int), ("plp_power", C.c_double),
                ("filter_type", C.c_int), ("boundary", C.c_int), ("warping", C.c_int)]

    @staticmethod
    def default(n_ceps=16, filter_width=268.258, sample_rate=16000.0, alpha=1.0):
        return MfccCfg(sample_rate, 0.025, 0.01, alpha, 0.025, 1, filter_width, 0.0, 1, n_ceps, 0, 0, 0, 0.33)

    @staticmethod
    def mfplp(n_ceps=13, n_autocorrelation=13, filter_width=268.258, sample_rate=16000.0, alpha=1.0):
        """mfplp.flow: nr-cepstrum-coefficients, nr-autocorrelation-coefficients (LPC order + 1)"""
        return MfccCfg(sample_rate, 0.025, 0.01, alpha, 0.025, 1, filter_width, 0.0, 1, n_ceps, 1, 1, n_autocorrelation, 0.33)

    @staticmethod
    def plp(n_ceps=13, n_autocorrelation=13, filter_width=3.8, spacing=0.93853, sample_rate=16000.0):
        """plp.flow: 20 ms Hamming window, no preemphasis, trapeze / include-boundary / bark filter bank, equal loudness"""
        return MfccCfg(sample_rate, 0.02, 0.01, 0.0, 0.02, 1, filter_width, spacing, 1, n_ceps, 1, 2, n_autocorrelation, 0.33,
                       1, 1, 1)


class _GmmModel(C.Structure):
    _fields_ = [("dim", C.c_int), ("n_mix", C.c_int), ("n_dens", C.c_int), ("n_mean", C.c_int), ("n_cov", C.c_int),
                ("mix_offsets", C.c_void_p), ("dens_index", C.c_void_p), ("log_weight", C.c_void_p),
                ("dens_mean", C.c_void_p), ("dens_cov", C.c_void_p), ("means", C.c_void_p),
                ("variances", C.c_void_p), ("mixture_weight_scale", C.c_double), ("gaussian_scale", C.c_double)]


class _FfnnModel(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("in_dim", C.c_void_p), ("out_dim", C.c_void_p), ("W", C.c_void_p),
                ("bias", C.c_void_p), ("activation", C.c_void_p), ("log_prior", C.c_void_p),
                ("prior_scale", C.c_float)]


def build_oracle(force=False):
    """gcc-compile oracle/liboracle.so (and _ref/libref.so when the reference tree is mounted)."""
    if force or any(not os.path.exists(lib) or any(os.path.getmtime(os.path.join(_HERE, s)) > os.path.getmtime(lib) for s in _SRCS + ("orc.h",))
                    for lib in _LIBS.values()):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and (force or not all(os.path.exists(r) for r in _REFS.values())):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "ref")], stdout=subprocess.DEVNULL)


_libs = {}
_NATIVE = {}     # contract -> path of the -march=native build (bench.py's cpu_baseline only), see build_native_oracle
_use_native = False
DEFAULT_CONTRACT = "off"


def build_native_oracle(contract="off"):
    """The same sources tuned for THIS host (-O3 -march=native), into a temporary directory: the file is host-specific and must not
    travel.  -ffp-contract=off in both modes -- the compiler fuses nothing on its own, so the results are those of the -O2 checker
    library of the same mode; contract "fma" adds -DORC_CONTRACT_FMA, and with -march=native its fmaf() sites become vfmadd
    instructions, i.e. the instruction mix of the reference's DEFAULT build (-march=native, GCC's -ffp-contract=fast).  Used by
    bench.py's cpu_baseline, never by the tests."""
    if _NATIVE.get(contract) and os.path.exists(_NATIVE[contract]):
        return _NATIVE[contract]
    import tempfile
    out = os.path.join(tempfile.gettempdir(), "liboracle_native_%s_%d.so" % (contract, os.getuid()))
    srcs = [os.path.join(_HERE, f) for f in _SRCS]
    if not os.path.exists(out) or any(os.path.getmtime(f) > os.path.getmtime(out) for f in srcs + [os.path.join(_HERE, "orc.h")]):
        tmp = out + ".%d.tmp" % os.getpid()
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-std=gnu11", "-w"] +
                              (["-DORC_CONTRACT_FMA"] if contract == "fma" else []) + ["-shared", "-o", tmp] + srcs + ["-lm"])
        os.replace(tmp, out)
    _NATIVE[contract] = out
    return out


def use_native_oracle():
    """make Oracle() of this process load the -march=native builds (call before the first Oracle())"""
    global _use_native
    _use_native = True
    _libs.clear()


def set_default_contract(contract):
    """the mode Oracle() and the Oracle* classes use when none is named (bench.py's cpu_baseline worker processes)"""
    global DEFAULT_CONTRACT
    assert contract in CONTRACTS, contract
    DEFAULT_CONTRACT = contract


def Oracle(contract=None):
    """the oracle library that restates the reference's `contract` build ("off" | "fma"; None = DEFAULT_CONTRACT)"""
    contract = DEFAULT_CONTRACT if contract is None else contract
    assert contract in CONTRACTS, contract
    if contract in _libs:
        return _libs[contract]
    if _use_native:
        path = build_native_oracle(contract)
    else:
        build_oracle()
        path = _LIBS[contract]
    L = C.CDLL(path)
    L.orc_contract.restype = C.c_int
    assert L.orc_contract() == (1 if contract == "fma" else 0), "%s restates the other build" % path
    L.orc_gmm_distance.restype = C.c_float
    L.orc_gmm_distance.argtypes = [f32p, f32p, f32p, C.c_int]
    L.orc_filter_apply.restype = C.c_float
    L.orc_filter_apply.argtypes = [f32p, C.c_int, C.c_int, f32p]
    L.orc_hamming_window.argtypes = [f32p, C.c_int]
    L.orc_cluster_u8.restype = None
    L.orc_cluster_u8.argtypes = [np.ctypeslib.ndpointer(np.uint8, flags="C"), C.c_int, C.c_int, C.c_int, C.c_int,
                                 np.ctypeslib.ndpointer(np.uint32, flags="C"), np.ctypeslib.ndpointer(np.uint8, flags="C")]
    L.orc_cluster_select.restype = None
    L.orc_cluster_select.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, np.ctypeslib.ndpointer(np.uint8, flags="C")]
    L.orc_window_value.restype = C.c_float
    L.orc_window_value.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_temporal_integrate.restype = None
    L.orc_temporal_integrate.argtypes = [C.c_int, f32p, C.c_int, C.c_int, f32p]
    L.orc_spectral_integrate.restype = C.c_int
    L.orc_spectral_integrate.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_int, f32p]
    L.orc_cosine_transform.restype = None
    L.orc_cosine_transform.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p]
    L.orc_filter_boundary.restype = C.c_int
    L.orc_filter_boundary.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), np.ctypeslib.ndpointer(np.float64, flags="C"), C.c_int]
    L.orc_filter_build.restype = C.c_int
    L.orc_filter_build.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int), f32p, C.c_int]
    L.orc_batch_float_fill.restype = C.c_float
    L.orc_batch_float_fill.argtypes = [f32p, f32p, C.c_int, f32p, C.c_int]
    L.orc_mfcc_create.restype = C.c_void_p
    L.orc_mfcc_create.argtypes = [C.POINTER(MfccCfg)]
    L.orc_mfcc_destroy.argtypes = [C.c_void_p]
    for n in ("frame_len", "frame_shift", "fft_len", "n_bins", "n_filters", "n_ceps"):
        f = getattr(L, "orc_mfcc_" + n)
        f.restype = C.c_int
        f.argtypes = [C.c_void_p]
    L.orc_mfcc_n_frames.restype = C.c_long
    L.orc_mfcc_n_frames.argtypes = [C.c_void_p, C.c_long]
    for n, t in (("window", C.c_float), ("filter_start", C.c_int), ("filter_end", C.c_int),
                 ("filter_offset", C.c_int), ("filter_weights", C.c_float), ("dct", C.c_float), ("equal_loudness", C.c_double)):
        f = getattr(L, "orc_mfcc_" + n)
        f.restype = C.POINTER(t)
        f.argtypes = [C.c_void_p]
    L.orc_mfcc_mel_max.restype = C.c_double
    L.orc_mfcc_mel_max.argtypes = [C.c_void_p]
    L.orc_mfcc_run.restype = C.c_long
    L.orc_mfcc_run.argtypes = [C.c_void_p, f32p, C.c_long, f32p]
    L.orc_mfcc_stages.restype = C.c_int
    L.orc_mfcc_stages.argtypes = [C.c_void_p, f32p, C.c_long, C.c_long] + [C.c_void_p] * 6
    L.orc_preemphasis.argtypes = [f32p, C.c_long, C.c_float]
    L.orc_fft_real.argtypes = [f32p, C.c_int]
    L.orc_fft_complex.argtypes = [f32p, C.c_int]
    for n in ("orc_core_is_almost_equal", "orc_core_is_significantly_greater"):
        getattr(L, n).restype = C.c_int
        getattr(L, n).argtypes = [C.c_double, C.c_double, C.c_double]
    for n in ("orc_mel", "orc_mel_derivative", "orc_mel_inverse", "orc_bark", "orc_bark_derivative", "orc_bark_inverse",
              "orc_equal_loudness", "orc_equal_loudness_4khz"):
        getattr(L, n).restype = C.c_double
        getattr(L, n).argtypes = [C.c_double]
    L.orc_gmm_create.restype = C.c_void_p
    L.orc_gmm_create.argtypes = [C.POINTER(_GmmModel)]
    L.orc_gmm_destroy.argtypes = [C.c_void_p]
    for n in ("minus2_log_weights", "inv_sqrt_var", "log_norm"):
        f = getattr(L, "orc_gmm_" + n)
        f.restype = C.POINTER(C.c_float)
        f.argtypes = [C.c_void_p]
    L.orc_gmm_score.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int, f32p, C.c_void_p]
    L.orc_gmm_score_batch_float.restype = C.c_int
    L.orc_gmm_score_batch_float.argtypes = [C.c_void_p, f64p, f32p, f32p, C.c_int, f32p]
    L.orc_gmm_score_simd.restype = C.c_int
    L.orc_gmm_score_simd.argtypes = [C.c_void_p, f64p, f32p, f32p, C.c_int, f32p, C.c_void_p, C.c_void_p]
    L.orc_quantize.restype = C.c_uint
    L.orc_quantize.argtypes = [C.c_float]
    L.orc_gmm_accumulator_size.restype = C.c_long
    L.orc_gmm_accumulator_size.argtypes = [C.c_void_p]
    L.orc_gmm_accumulate.argtypes = [C.c_void_p, f32p, C.c_int, u32p, u32p, f64p]
    L.orc_gmm_accumulate_weighted.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int, u32p, C.c_void_p, C.c_void_p, f64p]
    L.orc_ffnn_score.argtypes = [C.POINTER(_FfnnModel), f32p, C.c_int, f32p, C.c_int]
    _libs[contract] = L
    return L


_refs = {}


def load_ref(contract="off"):
    """oracle/_ref/libref.so (contract "off": reference TUs compiled unmodified with -msse3) or libref_native.so (contract "fma":
    the same with the reference's default -march=native, host-specific, build container only); None when not built."""
    if contract in _refs:
        return _refs[contract]
    path = _REFS[contract]
    if not os.path.exists(path):
        if os.path.isdir("/root/reference/src"):
            build_oracle()
        if not os.path.exists(path):
            return None
    R = C.CDLL(path)
    R.ref_fft_real.argtypes = [f32p, C.c_int]
    R.ref_fft_complex.argtypes = [f32p, C.c_int]
    for n in ("ref_mel", "ref_mel_derivative", "ref_mel_inverse", "ref_bark", "ref_bark_derivative", "ref_bark_inverse"):
        getattr(R, n).restype = C.c_double
        getattr(R, n).argtypes = [C.c_double]
    for n in ("ref_is_almost_equal", "ref_is_significantly_greater"):
        getattr(R, n).restype = C.c_int
        getattr(R, n).argtypes = [C.c_double, C.c_double, C.c_double]
    R.ref_equal_loudness.restype = C.c_double
    R.ref_equal_loudness.argtypes = [C.c_double, C.c_int]
    R.ref_plp_equal_loudness.restype = C.c_double
    R.ref_plp_equal_loudness.argtypes = [C.c_double, C.c_double, C.c_int]
    for n in ("ref_warped_bin", "ref_warped_bin_inverse", "ref_warped_bin_derivative", "ref_bark_bin", "ref_bark_bin_inverse",
              "ref_bark_bin_derivative"):
        getattr(R, n).restype = C.c_double
        getattr(R, n).argtypes = [C.c_double, C.c_double]
    R.ref_gauss_log_norm_factor.restype = C.c_double
    R.ref_gauss_log_norm_factor.argtypes = [f32p, C.c_int]
    R.ref_inverse_square_root.restype = C.c_float
    R.ref_inverse_square_root.argtypes = [C.c_float]
    R.ref_time_window_frames.restype = C.c_long
    R.ref_time_window_frames.argtypes = [C.c_long, C.c_long, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_double, C.c_long,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    R.ref_mt_vr_exp.restype = None
    R.ref_mt_vr_exp.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    R.ref_window_frames.restype = C.c_long
    R.ref_window_frames.argtypes = [f32p, C.c_long, C.c_long, C.c_uint, C.c_uint, C.c_double, C.c_long,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
    R.ref_cache_block_write.restype = C.c_long
    R.ref_cache_block_write.argtypes = [f32p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long]
    R.ref_cache_block_read.restype = C.c_long
    R.ref_cache_block_read.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_long,
                                       C.POINTER(C.c_long), C.c_char_p, C.c_int]
    R.ref_attribs_xml.restype = C.c_long
    R.ref_attribs_xml.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_long]
    if hasattr(R, "ref_matrix_vector"):
        R.ref_matrix_vector.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        R.ref_complex_amplitude.restype = C.c_int
        R.ref_complex_amplitude.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    if hasattr(R, "ref_gdm_score"):
        R.ref_gdm_score.restype = None
        R.ref_gdm_score.argtypes = [C.c_int, f32p, C.c_int, C.c_int, f32p, f32p, f32p, f32p, C.POINTER(C.c_float), C.POINTER(C.c_uint), C.c_void_p]
    if hasattr(R, "ref_gdm_distance"):   # function-text pins (oracle/ref/extract_fn.py)
        R.ref_gdm_distance.restype = C.c_float
        R.ref_gdm_distance.argtypes = [f32p, f32p, f32p, C.c_int]
        R.ref_regression.argtypes = [C.c_int, f32p, C.c_int, C.c_int, f32p]
        R.ref_filter_apply.restype = C.c_float
        R.ref_filter_apply.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p]
    if hasattr(R, "ref_hamming_window"):
        R.ref_hamming_window.restype = C.c_int
        R.ref_hamming_window.argtypes = [C.c_int, f32p]
    if hasattr(R, "ref_window_table"):
        R.ref_window_table.restype = C.c_int
        R.ref_window_table.argtypes = [C.c_int, C.c_int, f32p]
        R.ref_temporal_integration.restype = None
        R.ref_temporal_integration.argtypes = [C.c_int, f32p, C.c_int, C.c_int, f32p]
        R.ref_spectral_integration.restype = C.c_int
        R.ref_spectral_integration.argtypes = [C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int, f32p]
        R.ref_batch_float_fill.argtypes = [f32p, f32p, C.c_int, f32p, C.c_int, C.c_int, f32p]
    if hasattr(R, "ref_filter_build"):
        R.ref_filter_build.restype = C.c_int
        R.ref_filter_build.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), f32p, C.c_int]
    if hasattr(R, "ref_filter_boundary"):
        R.ref_filter_boundary.restype = C.c_int
        R.ref_filter_boundary.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), np.ctypeslib.ndpointer(np.float64, flags="C"), C.c_int]
    if hasattr(R, "ref_cosine_transform"):
        R.ref_cosine_transform.restype = None
        R.ref_cosine_transform.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p]
    if hasattr(R, "ref_ar_to_cepstrum"):
        R.ref_ar_to_cepstrum.restype = None
        R.ref_ar_to_cepstrum.argtypes = [C.c_float, f32p, C.c_int, f32p, C.c_int]
    if hasattr(R, "ref_gammatone"):
        R.ref_gammatone.restype = C.c_int
        R.ref_gammatone.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_char_p, f32p,
                                    C.c_long, C.c_int, f32p, f32p, f32p]
    if hasattr(R, "ref_normalization"):
        R.ref_normalization.restype = C.c_long
        R.ref_normalization.argtypes = [C.c_int, C.c_int, C.c_ulong, C.c_ulong, f32p, C.c_long, C.c_int, f32p]
    if hasattr(R, "ref_density_clustering"):
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
        R.ref_density_clustering.restype = C.c_int
        R.ref_density_clustering.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, f32p, f32p, C.c_int, u8p]
        R.ref_density_clustering_u8.restype = C.c_int
        R.ref_density_clustering_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p]
    if hasattr(R, "ref_vector_function"):
        R.ref_vector_function.restype = None
        R.ref_vector_function.argtypes = [C.c_int, C.c_float, f32p, C.c_long, f32p]
    if hasattr(R, "ref_vector_normalize"):
        R.ref_vector_normalize.restype = None
        R.ref_vector_normalize.argtypes = [C.c_int, f32p, C.c_int, f32p]
    if hasattr(R, "ref_preemphasis"):
        R.ref_preemphasis.argtypes = [C.c_float, C.c_double, f32p, C.c_long, C.c_int, C.c_int, f32p]
    _refs[contract] = R
    return R


def oracle_normalize(x, variance=False, length=0, right=0):
    """orc_normalize over one segment [n, dim]"""
    L = Oracle()
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    L.orc_normalize.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    L.orc_normalize(x.reshape(-1), x.shape[0], x.shape[1], 1 if variance else 0, length, right, out.reshape(-1))
    return out


def oracle_normalize_ex(x, type, level=0, length=0, right=0, contract=None):
    """orc_normalize_ex over one segment [n, dim]: type 2 divide-by-mean, 3 level, 4 mean-and-variance-1D (0 / 1 as oracle_normalize)"""
    L = Oracle(contract)
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    L.orc_normalize_ex.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    L.orc_normalize_ex(x.reshape(-1), x.shape[0], x.shape[1], type, level, length, right, out.reshape(-1))
    return out


def oracle_vector_normalize(x, kind, contract=None):
    """signal-vector-f32-<kind>-normalization of every row of x"""
    L = Oracle(contract)
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    types = {"amplitude-spectrum-energy": 0, "energy": 1, "maximum": 2, "mean-energy": 3, "mean": 4, "variance": 5}
    L.orc_vector_normalize.argtypes = [C.c_int, f32p, C.c_int, C.c_int, f32p]
    L.orc_vector_normalize(types[kind], x.reshape(-1), x.shape[0], x.shape[1], out.reshape(-1))
    return out


def oracle_regression(x, order=1, right=2, contract=None):
    L = Oracle(contract)
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    L.orc_regression.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    L.orc_regression(x.reshape(-1), x.shape[0], x.shape[1], order, right, out.reshape(-1))
    return out


def oracle_matrix_multiply(M, x, contract=None):
    L = Oracle(contract)
    M = np.ascontiguousarray(M, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty((x.shape[0], M.shape[0]), np.float32)
    L.orc_matrix_multiply.argtypes = [f32p, C.c_int, C.c_int, f32p, C.c_int, f32p]
    L.orc_matrix_multiply(M.reshape(-1), M.shape[0], M.shape[1], x.reshape(-1), x.shape[0], out.reshape(-1))
    return out


def ref_cache_block(feats, times):
    """bytes of one Flow cache block written by the reference's Flow::Vector<f32> / Datatype code (libref)"""
    R = load_ref()
    x = np.ascontiguousarray(feats, np.float32)
    t = np.ascontiguousarray(times, np.float64)
    cap = x.size * 4 + x.shape[0] * 20 + 64
    buf = C.create_string_buffer(cap)
    n = R.ref_cache_block_write(x, t.ctypes.data, x.shape[0], x.shape[1], buf, cap)
    if n < 0:
        raise RuntimeError("ref_cache_block_write failed (%d)" % n)
    return buf.raw[:n]


def ref_cache_block_parse(data, dim, max_frames):
    """parse one block with the reference reader -> (feats, times, consumed bytes, datatype name)"""
    R = load_ref()
    x = np.zeros((max_frames, dim), np.float32)
    t = np.zeros((max_frames, 2), np.float64)
    used, name = C.c_long(), C.create_string_buffer(64)
    n = R.ref_cache_block_read(data, len(data), dim, x.ctypes.data, t.ctypes.data, max_frames, C.byref(used), name, 64)
    if n < 0:
        raise RuntimeError("ref_cache_block_read failed (%d)" % n)
    return x[:n], t[:n], used.value, name.value.decode()


def ref_attribs_xml(attrs):
    R = load_ref()
    names = (C.c_char_p * len(attrs))(*[k.encode() for k in attrs])
    vals = (C.c_char_p * len(attrs))(*[str(v).encode() for v in attrs.values()])
    buf = C.create_string_buffer(1 << 16)
    n = R.ref_attribs_xml(names, vals, len(attrs), buf, 1 << 16)
    if n < 0:
        raise RuntimeError("ref_attribs_xml failed")
    return buf.value.decode()


def ref_accumulate_vector(sum_, v, weight, kind):
    """reference functors on one accumulator row (libref): kind 0 plus, 1 plusWeighted, 2 plusSquare, 3 plusSquareWeighted (v f32),
    4 plusNormalizedSquare (v f64)"""
    R = load_ref()
    s = np.array(sum_, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64 if kind == 4 else np.float32)
    R.ref_accumulate_vector.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int]
    R.ref_accumulate_vector.restype = None
    R.ref_accumulate_vector(s.ctypes.data, v.ctypes.data, len(s), float(weight), kind)
    return s


def ref_log_exp_norm(v):
    R = load_ref()
    v = np.ascontiguousarray(v, dtype=np.float64)
    R.ref_log_exp_norm.argtypes = [C.c_void_p, C.c_int]
    R.ref_log_exp_norm.restype = C.c_double
    return R.ref_log_exp_norm(v.ctypes.data, len(v))


def ref_normalized_minus(x, y, weight):
    R = load_ref()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.zeros(len(x), np.float32)
    R.ref_normalized_minus.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    R.ref_normalized_minus.restype = None
    R.ref_normalized_minus(x.ctypes.data, y.ctypes.data, len(x), float(weight), out.ctypes.data)
    return out


class OracleMfcc:
    def __init__(self, cfg=None, contract=None, **kw):
        self.L = Oracle(contract)
        self.cfg = cfg if cfg is not None else MfccCfg.default(**kw)
        self.h = self.L.orc_mfcc_create(C.byref(self.cfg))
        if not self.h:
            raise ValueError("oracle: invalid MFCC configuration")
        g = lambda n: getattr(self.L, "orc_mfcc_" + n)(self.h)
        self.frame_len, self.frame_shift, self.fft_len = g("frame_len"), g("frame_shift"), g("fft_len")
        self.n_bins, self.n_filters, self.n_ceps = g("n_bins"), g("n_filters"), g("n_ceps")
        self.mel_max = self.L.orc_mfcc_mel_max(self.h)

    def __del__(self):
        try:
            self.L.orc_mfcc_destroy(self.h)
        except Exception:
            pass

    def _arr(self, name, n, dt):
        p = getattr(self.L, "orc_mfcc_" + name)(self.h)
        return np.ctypeslib.as_array(p, shape=(n,)).astype(dt).copy()

    @property
    def window(self):
        return self._arr("window", self.frame_len, np.float32)

    @property
    def filters(self):
        s = self._arr("filter_start", self.n_filters, np.int32)
        e = self._arr("filter_end", self.n_filters, np.int32)
        o = self._arr("filter_offset", self.n_filters + 1, np.int32)
        w = self._arr("filter_weights", int(o[-1]), np.float32)
        return s, e, o, w

    @property
    def dct(self):
        rows = self.cfg.n_autocorrelation if self.cfg.front_end != 0 else self.n_ceps
        cols = self.n_transform_inputs
        return self._arr("dct", rows * cols, np.float32).reshape(rows, cols)

    @property
    def n_transform_inputs(self):
        """size of the vector the cosine transform sees: the filter-bank outputs, plus the duplicated first / last one (plp.flow)"""
        return self.n_filters + (2 if self.cfg.front_end == 2 else 0)

    @property
    def equal_loudness(self):
        return self._arr("equal_loudness", self.n_transform_inputs, np.float64) if self.cfg.front_end == 2 else None

    def n_frames(self, n):
        return int(self.L.orc_mfcc_n_frames(self.h, n))

    def run(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        T = self.n_frames(len(pcm))
        out = np.zeros((T, self.n_ceps), dtype=np.float32)
        if T:
            self.L.orc_mfcc_run(self.h, pcm, len(pcm), out.reshape(-1))
        return out

    def stages(self, pcm, frame):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        o = dict(windowed=np.zeros(self.fft_len, np.float32), spectrum=np.zeros(self.fft_len + 2, np.float32),
                 amplitude=np.zeros(self.n_bins, np.float32), mel=np.zeros(self.n_filters, np.float32),
                 logmel=np.zeros(self.n_transform_inputs, np.float32), ceps=np.zeros(self.n_ceps, np.float32))
        r = self.L.orc_mfcc_stages(self.h, pcm, len(pcm), frame, *[v.ctypes.data for v in o.values()])
        if r != 0:
            raise IndexError("frame out of range")
        return o


class OracleGmm:
    """model: dict with dim, mix_offsets(u32), dens_index(u32), log_weight(f64), dens_mean, dens_cov (u32),
    means [n_mean,dim] f32, variances [n_cov,dim] f32."""

    def __init__(self, model, mixture_weight_scale=1.0, gaussian_scale=1.0, contract=None):
        """contract: which build of the reference to restate -- "off" (-DMARCH=x86-64: no fused multiply-add) or "fma" (its default
        -march=native build on an FMA host: `sum += df * df` of the distance is one fused operation); None = the module default"""
        self.L = Oracle(contract)
        self.m = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in model.items()}
        m = self.m
        self.n_mix = len(m["mix_offsets"]) - 1
        self.dim = int(m["dim"])
        st = _GmmModel(self.dim, self.n_mix, len(m["dens_mean"]), m["means"].shape[0], m["variances"].shape[0],
                       m["mix_offsets"].ctypes.data, m["dens_index"].ctypes.data, m["log_weight"].ctypes.data,
                       m["dens_mean"].ctypes.data, m["dens_cov"].ctypes.data, m["means"].ctypes.data,
                       m["variances"].ctypes.data, mixture_weight_scale, gaussian_scale)
        assert m["mix_offsets"].dtype == np.uint32 and m["log_weight"].dtype == np.float64
        assert m["means"].dtype == np.float32 and m["variances"].dtype == np.float32
        self.h = self.L.orc_gmm_create(C.byref(st))

    def __del__(self):
        try:
            self.L.orc_gmm_destroy(self.h)
        except Exception:
            pass

    def tables(self):
        nk = int(self.m["mix_offsets"][-1])
        nc = self.m["variances"].shape[0]
        a = lambda n, k: np.ctypeslib.as_array(getattr(self.L, "orc_gmm_" + n)(self.h), shape=(k,)).copy()
        return a("minus2_log_weights", nk), a("inv_sqrt_var", nc * self.dim).reshape(nc, self.dim), a("log_norm", nc)

    def score(self, feats, mode=0, want_best=True):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        T = feats.shape[0]
        sc = np.zeros((T, self.n_mix), np.float32)
        best = np.zeros((T, self.n_mix), np.uint32) if want_best else None
        self.L.orc_gmm_score(self.h, mode, feats.reshape(-1), T, sc.reshape(-1),
                             best.ctypes.data if want_best else None)
        return (sc, best) if want_best else sc

    def accumulator_size(self):
        return int(self.L.orc_gmm_accumulator_size(self.h))

    def accumulate(self, feats, mixture, density_in_mixture, acc=None):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        if acc is None:
            acc = np.zeros(self.accumulator_size(), np.float64)
        self.L.orc_gmm_accumulate(self.h, feats.reshape(-1), feats.shape[0], np.ascontiguousarray(mixture, dtype=np.uint32),
                                  np.ascontiguousarray(density_in_mixture, dtype=np.uint32), acc)
        return acc

    def accumulate_weighted(self, mode, feats, mixture, weight=None, density_in_mixture=None, acc=None):
        """mode 0: weighted Viterbi (needs density_in_mixture); mode 1: Baum-Welch posteriors of the log-add scorer"""
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        if acc is None:
            acc = np.zeros(self.accumulator_size(), np.float64)
        w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
        b = None if density_in_mixture is None else np.ascontiguousarray(density_in_mixture, dtype=np.uint32)
        self.L.orc_gmm_accumulate_weighted(self.h, mode, feats.reshape(-1), feats.shape[0], np.ascontiguousarray(mixture, dtype=np.uint32),
                                           None if w is None else w.ctypes.data, None if b is None else b.ctypes.data, acc)
        return acc

    def score_batch_float(self, feats):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        T = feats.shape[0]
        sc = np.zeros((T, self.n_mix), np.float32)
        r = self.L.orc_gmm_score_batch_float(self.h, self.m["log_weight"], self.m["variances"].reshape(-1),
                                             feats.reshape(-1), T, sc.reshape(-1))
        if r != 0:
            raise ValueError("batch-float scorer supports only a globally pooled covariance")
        return sc


def _preselection(self, feats, n_clusters=256, n_select=32, iterations=5, backoff=40000.0):
    """preselection-batch-float: (scores, cluster index per mixture entry, cluster means [n_clusters, padded dim])"""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    T = feats.shape[0]
    nk = int(self.m["mix_offsets"][-1])
    pdim = (self.dim + 7) // 8 * 8
    sc = np.zeros((T, self.n_mix), np.float32)
    cof = np.zeros(nk, np.uint32)
    cm = np.zeros((min(n_clusters, nk), pdim), np.float32)
    nc = C.c_int(0)
    self.L.orc_gmm_score_preselection_float.restype = C.c_int
    self.L.orc_gmm_score_preselection_float.argtypes = [C.c_void_p, f64p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, f32p,
                                                        u32p, f32p, C.POINTER(C.c_int)]
    r = self.L.orc_gmm_score_preselection_float(self.h, self.m["log_weight"], self.m["variances"].reshape(-1), feats.reshape(-1), T,
                                                n_clusters, n_select, iterations, backoff, sc.reshape(-1), cof, cm.reshape(-1), C.byref(nc))
    if r != 0:
        raise ValueError("preselection scorer: pooled covariance only, 1 <= select-clusters <= clusters (status %d)" % r)
    return sc, cof, cm


OracleGmm.score_preselection_float = _preselection


def _preselection_int(self, feats, n_clusters=256, n_select=32, iterations=5):
    """preselection-batch-int: (scores, cluster index per mixture entry, cluster means [n_clusters, dim] u8)"""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    T = feats.shape[0]
    nk = int(self.m["mix_offsets"][-1])
    sc = np.zeros((T, self.n_mix), np.float32)
    cof = np.zeros(nk, np.uint32)
    cm = np.zeros((min(n_clusters, nk), self.dim), np.uint8)
    nc = C.c_int(0)
    self.L.orc_gmm_score_preselection_int.restype = C.c_int
    self.L.orc_gmm_score_preselection_int.argtypes = [C.c_void_p, f64p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, u32p,
                                                      C.c_void_p, C.POINTER(C.c_int)]
    r = self.L.orc_gmm_score_preselection_int(self.h, self.m["log_weight"], self.m["variances"].reshape(-1), feats.reshape(-1), T,
                                              n_clusters, n_select, iterations, sc.reshape(-1), cof, cm.ctypes.data, C.byref(nc))
    if r != 0:
        raise ValueError("preselection-int scorer: pooled covariance only, 1 <= select-clusters <= clusters (status %d)" % r)
    return sc, cof, cm


OracleGmm.score_preselection_int = _preselection_int


def _simd(self, feats):
    """SIMD-diagonal-maximum: (scores, best density, scaling)"""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    T = feats.shape[0]
    sc = np.zeros((T, self.n_mix), np.float32)
    best = np.zeros((T, self.n_mix), np.uint32)
    scaling = C.c_float()
    self.L.orc_gmm_score_simd(self.h, self.m["log_weight"], self.m["variances"].reshape(-1), feats.reshape(-1), T, sc.reshape(-1),
                              best.ctypes.data, C.addressof(scaling))
    return sc, best, scaling.value


OracleGmm.score_simd = _simd


def _batch_int(self, feats):
    """batch-diagonal-maximum-int / -fast scores"""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    T = feats.shape[0]
    sc = np.zeros((T, self.n_mix), np.float32)
    self.L.orc_gmm_score_batch_int.restype = C.c_int
    self.L.orc_gmm_score_batch_int.argtypes = [C.c_void_p, f64p, f32p, f32p, C.c_int, f32p]
    if self.L.orc_gmm_score_batch_int(self.h, self.m["log_weight"], self.m["variances"].reshape(-1), feats.reshape(-1), T, sc.reshape(-1)) != 0:
        raise ValueError("batch-int scorer supports only a globally pooled covariance")
    return sc


OracleGmm.score_batch_int = _batch_int


def _levinson(fn, R):
    R = np.ascontiguousarray(R, dtype=np.float32)
    gain = C.c_float()
    a = np.zeros(max(len(R) - 1, 1), np.float32)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    ok = fn(R.ctypes.data, len(R), C.addressof(gain), a.ctypes.data)
    return (np.float32(gain.value), a[:len(R) - 1]) if ok else None


def oracle_levinson(R):
    """(gain, a) or None: Math::LevinsonLeastSquares restated in oracle/orc_mfcc.c"""
    return _levinson(Oracle().orc_levinson, R)


def ref_levinson(R):
    """the reference's own Math/LevinsonLse.cc (libref)"""
    return _levinson(load_ref().ref_levinson, R)


def oracle_ar_to_cepstrum(gain, a, nc, contract=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    c = np.zeros(nc, np.float32)
    L = Oracle(contract)
    L.orc_ar_to_cepstrum.restype = None
    L.orc_ar_to_cepstrum.argtypes = [C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.orc_ar_to_cepstrum(float(gain), a.ctypes.data, len(a), c.ctypes.data, nc)
    return c


def oracle_quantize(v):
    return int(Oracle().orc_quantize(float(np.float32(v))))


def ref_quantize(v):
    """the reference's quantize<f32, u8> functor (Mm/Utilities.hh:190-202) through libref"""
    R = load_ref()
    R.ref_quantize_u8.restype = C.c_uint
    R.ref_quantize_u8.argtypes = [C.c_float]
    return int(R.ref_quantize_u8(float(np.float32(v))))


def oracle_ffnn_score(Ws, biases, acts, feats, log_prior=None, prior_scale=1.0, acc64=False, top=None):
    """Ws[l]: [out,in] f32; returns scores [T,out_last] = -(Wx+b-alpha*logprior)."""
    L = Oracle()
    n = len(Ws)
    Ws = [np.ascontiguousarray(w, dtype=np.float32) for w in Ws]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    ind = np.array([w.shape[1] for w in Ws], np.int32)
    outd = np.array([w.shape[0] for w in Ws], np.int32)
    act = np.array(acts, np.int32)
    Wp = (C.c_void_p * n)(*[w.ctypes.data for w in Ws])
    Bp = (C.c_void_p * n)(*[b.ctypes.data for b in bs])
    lp = None if log_prior is None else np.ascontiguousarray(log_prior, dtype=np.float32)
    st = _FfnnModel(n, ind.ctypes.data, outd.ctypes.data, C.cast(Wp, C.c_void_p), C.cast(Bp, C.c_void_p),
                    act.ctypes.data, None if lp is None else lp.ctypes.data, prior_scale)
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    T = feats.shape[0]
    out = np.zeros((T, int(outd[-1])), np.float32)
    if top is None:
        L.orc_ffnn_score(C.byref(st), feats.reshape(-1), T, out.reshape(-1), int(acc64))
    else:
        L.orc_ffnn_forward.argtypes = [C.c_void_p, f32p, C.c_int, f32p, C.c_int, C.c_int]
        L.orc_ffnn_forward(C.byref(st), feats.reshape(-1), T, out.reshape(-1), int(top), int(acc64))
    return out


def oracle_ffnn_forward(Ws, biases, acts, feats, top, log_prior=None, prior_scale=1.0, acc64=False):
    """the forward node's output [T, out_last]: top 0 = W x + b - alpha log prior, 1 = its softmax (orc_ffnn_forward)"""
    return oracle_ffnn_score(Ws, biases, acts, feats, log_prior, prior_scale, acc64, top=top)


def oracle_softmax_rows(x):
    """Math::FastMatrix<f32>::softmax per row (orc_softmax_rows)"""
    L = Oracle()
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    L.orc_softmax_rows.argtypes = [f32p, C.c_int, C.c_int]
    L.orc_softmax_rows(y.reshape(-1), y.shape[0], y.shape[1])
    return y


class GammatoneCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_double), ("cascade", C.c_int), ("min_freq", C.c_double), ("max_freq", C.c_double), ("q", C.c_double),
                ("channels", C.c_int), ("cf_mode", C.c_int), ("warp_freq_break", C.c_double), ("warping_factor", C.c_double),
                ("ti_window", C.c_int), ("ti_length_s", C.c_double), ("ti_shift_s", C.c_double), ("si_window", C.c_int),
                ("si_length", C.c_int), ("si_shift", C.c_int), ("power", C.c_double), ("n_ceps", C.c_int), ("dct_normalize", C.c_int)]

    @staticmethod
    def default(**kw):
        """the nodes' own defaults (signal-gammatone: cascade 4, 100..6000 Hz, 50 channels, human centre frequencies) with a 25 ms /
        10 ms Hanning temporal integration; spectral integration, root compression and cosine transform absent"""
        c = GammatoneCfg(16000.0, 4, 100.0, 6000.0, 9.264491981582191, 50, 0, 6600.0, 1.0, 0, 0.025, 0.01, 0, 0, 1, 0.0, 0, 0)
        for k, v in kw.items():
            setattr(c, k, v)
        return c


def oracle_activation(x, act):
    """orc_activation elementwise; act: 1 ReLU, 2 sigmoid, 3 tanh (the layers' own functions)"""
    L = Oracle()
    L.orc_activation.restype, L.orc_activation.argtypes = C.c_float, [C.c_float, C.c_int]
    return np.array([L.orc_activation(float(v), act) for v in np.asarray(x, np.float32).ravel()], np.float32)


def oracle_vector_function(x, kind, parameter=0.0):
    """generic-vector-f32-<function> elementwise (orc_vector_function); kind: index of AMX_VFUNC_*"""
    L = Oracle()
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros_like(x)
    L.orc_vector_function.argtypes = [C.c_int, C.c_float, f32p, C.c_long, C.c_int, f32p]
    L.orc_vector_function(int(kind), float(parameter), x.reshape(-1), x.shape[0], x.shape[1], y.reshape(-1))
    return y


def oracle_dc_detection(pcm, block=4096, sample_rate=16000.0, min_dc_length=0.0125, max_dc_increment=0.9, min_non_dc_segment_length=0.02,
                        maximal_output_size=4096):
    """signal-dc-detection: [(first sample, length)] of the vectors the node emits for one segment fed in vectors of `block` samples"""
    L = Oracle()
    x = np.ascontiguousarray(pcm, dtype=np.float32)
    L.orc_dc_detection.restype = C.c_longlong
    L.orc_dc_detection.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_double, C.c_double, C.c_float, C.c_double, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_longlong]
    args = (x.ctypes.data if len(x) else None, len(x), block, sample_rate, min_dc_length, max_dc_increment, min_non_dc_segment_length,
            maximal_output_size)
    n = L.orc_dc_detection(*args, None, None, 0)
    st, ln = np.zeros(n, np.int64), np.zeros(n, np.int64)
    L.orc_dc_detection(*args, st.ctypes.data, ln.ctypes.data, n)
    return list(zip(st.tolist(), ln.tolist()))


def oracle_time_window_frames(n, length, shift):
    """(starts, lens) of the frames signal-temporalintegration cuts out of n samples (orc_time_window_frames)"""
    L = Oracle()
    L.orc_time_window_frames.restype = C.c_long
    L.orc_time_window_frames.argtypes = [C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
    T = L.orc_time_window_frames(n, length, shift, None, None, 0)
    starts, lens = np.zeros(T, np.int64), np.zeros(T, np.int32)
    L.orc_time_window_frames(n, length, shift, starts.ctypes.data, lens.ctypes.data, T)
    return starts, lens


class OracleGammatone:
    def __init__(self, cfg=None, contract=None, **kw):
        self.L = Oracle(contract)
        L = self.L
        L.orc_gammatone_create.restype = C.c_void_p
        L.orc_gammatone_create.argtypes = [C.POINTER(GammatoneCfg)]
        L.orc_gammatone_destroy.argtypes = [C.c_void_p]
        for n in ("n_out", "frame_len", "frame_shift", "si_channels"):
            f = getattr(L, "orc_gammatone_" + n)
            f.restype, f.argtypes = C.c_int, [C.c_void_p]
        for n in ("center_frequencies", "coefficients"):
            f = getattr(L, "orc_gammatone_" + n)
            f.restype, f.argtypes = C.POINTER(C.c_float), [C.c_void_p]
        L.orc_gammatone_n_frames.restype, L.orc_gammatone_n_frames.argtypes = C.c_long, [C.c_void_p, C.c_long]
        L.orc_gammatone_run.restype = C.c_long
        L.orc_gammatone_run.argtypes = [C.c_void_p, f32p, C.c_long, C.c_void_p, f32p]
        self.cfg = cfg if cfg is not None else GammatoneCfg.default(**kw)
        self.h = L.orc_gammatone_create(C.byref(self.cfg))
        if not self.h:
            raise ValueError("oracle: invalid gammatone configuration")
        self.n_out, self.frame_len = L.orc_gammatone_n_out(self.h), L.orc_gammatone_frame_len(self.h)
        self.frame_shift, self.si_channels = L.orc_gammatone_frame_shift(self.h), L.orc_gammatone_si_channels(self.h)
        ch = self.cfg.channels
        self.center_frequencies = np.ctypeslib.as_array(L.orc_gammatone_center_frequencies(self.h), shape=(ch,)).copy()
        self.coefficients = np.ctypeslib.as_array(L.orc_gammatone_coefficients(self.h), shape=(ch, 4)).copy()

    def __del__(self):
        try:
            self.L.orc_gammatone_destroy(self.h)
        except Exception:
            pass

    def n_frames(self, n):
        return int(self.L.orc_gammatone_n_frames(self.h, n))

    def run(self, pcm, want_filtered=False):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        T = self.n_frames(len(pcm))
        out = np.zeros((T, self.n_out), np.float32)
        filt = np.zeros((len(pcm), self.cfg.channels), np.float32) if want_filtered else None
        if T:
            self.L.orc_gammatone_run(self.h, pcm, len(pcm), filt.ctypes.data if want_filtered else None, out.reshape(-1))
        return (out, filt) if want_filtered else out
