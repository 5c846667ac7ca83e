"""ctypes view of librasr_amd.so (include/amx.h).  No compute happens in Python.

The library is built in-tree by ``__graft_entry__.build()`` (``make -C rasr_amd/csrc``).  There is
no fallback: if the shared object is missing or no gfx950 device is visible, the calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMX_LIBRARY", os.path.join(_HERE, "librasr_amd.so"))  # AMX_LIBRARY: A/B runs of two builds (tools/)

AMX_OK, AMX_ERR_INVALID, AMX_ERR_UNSUPPORTED, AMX_ERR_DEVICE, AMX_ERR_STATE = 0, -1, -2, -3, -4
AMX_GMM_MAX, AMX_GMM_SUM, AMX_GMM_BATCH_FLOAT, AMX_GMM_SIMD, AMX_GMM_BATCH_INT, AMX_GMM_PRESELECTION_FLOAT, AMX_GMM_PRESELECTION_INT = 0, 1, 2, 3, 4, 5, 6
AMX_GMM_VITERBI, AMX_GMM_BAUM_WELCH = 0, 1
AMX_ACT_NONE, AMX_ACT_RELU, AMX_ACT_SIGMOID, AMX_ACT_TANH = 0, 1, 2, 3
AMX_PREC_FP32, AMX_PREC_BF16, AMX_PREC_BF16X3, AMX_PREC_F16MX = 0, 1, 2, 3
AMX_NN_TOP_LINEAR, AMX_NN_TOP_SOFTMAX = 0, 1
AMX_ARCHIVE_READ, AMX_ARCHIVE_WRITE = 0, 1
AMX_NORM_MEAN, AMX_NORM_MEAN_AND_VARIANCE = 0, 1


class AmxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("amx status %d: %s" % (status, message))
        self.status = status


class MfccCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_double), ("win_len_s", C.c_double), ("win_shift_s", C.c_double),
                ("preemph_alpha", C.c_double), ("fft_max_input_s", C.c_double), ("apply_scale", C.c_int),
                ("mel_filter_width", C.c_double), ("mel_spacing", C.c_double),
                ("warp_differential_unit", C.c_int), ("n_ceps", C.c_int), ("dct_normalize", C.c_int),
                ("front_end", C.c_int), ("n_autocorrelation", C.c_int), ("plp_power", C.c_double),
                ("filter_type", C.c_int), ("boundary", C.c_int), ("warping", C.c_int), ("tuning", C.c_char_p)]


class GammatoneCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_double), ("cascade", C.c_int), ("min_freq", C.c_double), ("max_freq", C.c_double), ("q", C.c_double),
                ("channels", C.c_int), ("cf_mode", C.c_int), ("warp_freq_break", C.c_double), ("warping_factor", C.c_double),
                ("ti_window", C.c_int), ("ti_length_s", C.c_double), ("ti_shift_s", C.c_double), ("si_window", C.c_int),
                ("si_length", C.c_int), ("si_shift", C.c_int), ("power", C.c_double), ("n_ceps", C.c_int), ("dct_normalize", C.c_int),
                ("tuning", C.c_char_p)]


class GammatoneInfo(C.Structure):
    _fields_ = [("channels", C.c_int), ("cascade", C.c_int), ("frame_len", C.c_int), ("frame_shift", C.c_int), ("si_channels", C.c_int),
                ("n_out", C.c_int)]


class MfccInfo(C.Structure):
    _fields_ = [("frame_len", C.c_int), ("frame_shift", C.c_int), ("fft_len", C.c_int), ("n_bins", C.c_int),
                ("n_filters", C.c_int), ("n_ceps", C.c_int), ("fft_output_sample_rate", C.c_double),
                ("mel_max", C.c_double), ("n_transform", C.c_int), ("n_transform_inputs", C.c_int)]


class GmmModel(C.Structure):
    _fields_ = [("dim", C.c_int), ("n_mix", C.c_int), ("n_dens", C.c_int), ("n_mean", C.c_int), ("n_cov", C.c_int),
                ("mix_offsets", C.c_void_p), ("dens_index", C.c_void_p), ("log_weight", C.c_void_p),
                ("dens_mean", C.c_void_p), ("dens_cov", C.c_void_p), ("means", C.c_void_p),
                ("variances", C.c_void_p), ("mixture_weight_scale", C.c_double), ("gaussian_scale", C.c_double), ("tuning", C.c_char_p)]


class GmmEstimateCfg(C.Structure):
    _fields_ = [("min_observation_weight", C.c_double), ("min_relative_weight", C.c_double), ("min_variance", C.c_double),
                ("normalize_mixture_weights", C.c_int), ("allow_zero_weights", C.c_int), ("split", C.c_int),
                ("split_min_mean_observation_weight", C.c_double), ("split_min_covariance_observation_weight", C.c_double),
                ("split_perturbation_weight", C.c_double), ("split_normalize_mixture_weights", C.c_int)]


class FfnnModel(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("in_dim", C.c_void_p), ("out_dim", C.c_void_p), ("W", C.c_void_p),
                ("bias", C.c_void_p), ("activation", C.c_void_p), ("log_prior", C.c_void_p),
                ("prior_scale", C.c_float), ("precision", C.c_int), ("n_classes", C.c_int), ("class_to_output", C.c_void_p),
                ("tuning", C.c_char_p)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against amx.h
_P = C.c_void_p
SIGNATURES = {
    "amx_version": (C.c_char_p, []),
    "amx_last_error": (C.c_char_p, []),
    "amx_init": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "amx_destroy": (None, [_P]),
    "amx_set_stream": (C.c_int, [_P, _P]),
    "amx_set_contract": (C.c_int, [_P, C.c_int]),
    "amx_get_contract": (C.c_int, [_P]),
    "amx_contract_description": (C.c_char_p, [C.c_int]),
    "amx_synchronize": (C.c_int, [_P]),
    "amx_profile_enable": (C.c_int, [_P, C.c_int]),
    "amx_profile_reset": (C.c_int, [_P]),
    "amx_profile_get": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "amx_mfcc_default_cfg": (None, [C.POINTER(MfccCfg)]),
    "amx_mfplp_default_cfg": (None, [C.POINTER(MfccCfg)]),
    "amx_plp_default_cfg": (None, [C.POINTER(MfccCfg)]),
    "amx_gammatone_default_cfg": (None, [C.POINTER(GammatoneCfg)]),
    "amx_gammatone_create": (C.c_int, [_P, C.POINTER(GammatoneCfg), C.POINTER(C.c_void_p)]),
    "amx_gammatone_destroy": (None, [_P]),
    "amx_gammatone_describe": (C.c_int, [_P, C.POINTER(GammatoneInfo)]),
    "amx_gammatone_n_frames": (C.c_long, [_P, C.c_long]),
    "amx_gammatone_tables": (C.c_int, [_P, _P, _P]),
    "amx_gammatone_run": (C.c_int, [_P, _P, C.c_long, _P]),
    "amx_gammatone_run_batch_dev": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "amx_mfcc_equal_loudness": (C.c_int, [_P, _P]),
    "amx_mfcc_create": (C.c_int, [_P, C.POINTER(MfccCfg), C.POINTER(_P)]),
    "amx_mfcc_destroy": (None, [_P]),
    "amx_mfcc_describe": (C.c_int, [_P, C.POINTER(MfccInfo)]),
    "amx_mfcc_n_frames": (C.c_long, [_P, C.c_long]),
    "amx_mfcc_frame_start_time": (C.c_double, [_P, C.c_long]),
    "amx_mfcc_tables": (C.c_int, [_P] * 7),
    "amx_mfcc_run": (C.c_int, [_P, _P, C.c_long, _P]),
    "amx_mfcc_run_batch": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "amx_mfcc_run_s16": (C.c_int, [_P, _P, C.c_long, _P]),
    "amx_mfcc_run_batch_s16": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "amx_mfcc_plan_create": (C.c_int, [_P, C.c_int, _P, C.POINTER(_P)]),
    "amx_mfcc_plan_destroy": (None, [_P]),
    "amx_mfcc_plan_total_frames": (C.c_long, [_P]),
    "amx_mfcc_plan_frame_offsets": (C.c_int, [_P, _P]),
    "amx_mfcc_run_plan_dev": (C.c_int, [_P, _P, _P, _P]),
    "amx_mfcc_run_plan_dev_s16": (C.c_int, [_P, _P, _P, _P]),
    "amx_context_window_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int]),
    "amx_normalize_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int]),
    "amx_normalize_ex_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int]),
    "amx_vector_normalize_dev": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_long, C.c_int, _P, C.c_int]),
    "amx_vector_function_dev": (C.c_int, [_P, C.c_int, C.c_float, _P, C.c_int, C.c_long, C.c_int, _P, C.c_int]),
    "amx_regression_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int]),
    "amx_matrix_multiply_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int]),
    "amx_gmm_create": (C.c_int, [_P, C.POINTER(GmmModel), C.POINTER(_P)]),
    "amx_gmm_destroy": (None, [_P]),
    "amx_gmm_n_mixtures": (C.c_int, [_P]),
    "amx_gmm_dimension": (C.c_int, [_P]),
    "amx_gmm_tables": (C.c_int, [_P, _P, _P, _P]),
    "amx_gmm_simd_scaling": (C.c_float, [_P]),
    "amx_gmm_set_preselection": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float]),
    "amx_gmm_preselection_clustering": (C.c_int, [_P, C.POINTER(C.c_int), _P, _P]),
    "amx_gmm_preselection_int_clustering": (C.c_int, [_P, C.POINTER(C.c_int), _P, _P]),
    "amx_gmm_screen_counts": (C.c_int, [_P, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "amx_gmm_score": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P]),
    "amx_gmm_score_dev": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P]),
    "amx_gmm_score_stats_dev": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P]),
    "amx_gmm_score_stats_u8_dev": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P]),
    "amx_gmm_best_density_dev": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "amx_gmm_accumulator_size": (C.c_long, [_P]),
    "amx_gmm_accumulate_u8_dev": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, _P]),
    "amx_gmm_accumulate_dev": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, _P]),
    "amx_gmm_accumulate_weighted_dev": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int, _P]),
    "amx_gmm_estimate_cfg_default": (None, [C.POINTER(GmmEstimateCfg)]),
    "amx_gmm_estimate": (C.c_int, [C.POINTER(GmmModel), _P, C.POINTER(GmmEstimateCfg), C.POINTER(_P)]),
    "amx_gmm_accumulator_write": (C.c_int, [_P, _P, C.c_char_p]),
    "amx_gmm_accumulator_read": (C.c_int, [_P, C.c_char_p, _P]),
    "amx_pms_read": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "amx_pms_write": (C.c_int, [C.POINTER(GmmModel), C.c_char_p]),
    "amx_mixture_set_view": (C.c_int, [_P, C.POINTER(GmmModel)]),
    "amx_mixture_set_destroy": (None, [_P]),
    "amx_ffnn_create": (C.c_int, [_P, C.POINTER(FfnnModel), C.POINTER(_P)]),
    "amx_ffnn_destroy": (None, [_P]),
    "amx_ffnn_input_dim": (C.c_int, [_P]),
    "amx_ffnn_output_dim": (C.c_int, [_P]),
    "amx_ffnn_score": (C.c_int, [_P, _P, C.c_int, _P]),
    "amx_ffnn_score_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "amx_device_clocks_dev": (C.c_int, [_P, _P]),
    "amx_device_clocks_xcd_dev": (C.c_int, [_P, _P]),
    "amx_ffnn_wait_dev": (C.c_int, [_P]),
    "amx_ffnn_precision": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "amx_ffnn_score_stats_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "amx_ffnn_hidden_dim": (C.c_int, [_P]),
    "amx_dc_detection": (C.c_int, [_P, C.c_longlong, C.c_double, C.c_double, C.c_float, C.c_double, C.c_int, C.c_int, _P, _P, C.c_longlong, _P]),
    "amx_ffnn_forward_hidden_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "amx_ffnn_forward_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int]),
    "amx_ffnn_score_on_demand_dev": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "amx_precomputed_score_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_float, _P]),
    "amx_class_labels_init": (C.c_int, [C.c_int, _P, C.c_int, _P, C.POINTER(C.c_int)]),
    "amx_nn_vector_read_f32": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(_P)]),
    "amx_nn_vector_write_f32": (C.c_int, [C.c_char_p, C.c_int, _P]),
    "amx_nn_vector_read_s32": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(_P)]),
    "amx_nn_vector_write_s32": (C.c_int, [C.c_char_p, C.c_int, _P]),
    "amx_nn_matrix_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_P)]),
    "amx_nn_matrix_write": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _P]),
    "amx_free": (None, [_P]),
    "amx_nn_layer_from_parameters": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "amx_prior_from_mixture_set": (C.c_int, [C.POINTER(GmmModel), _P]),
    "amx_archive_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "amx_archive_close": (C.c_int, [_P]),
    "amx_archive_n_files": (C.c_int, [_P]),
    "amx_archive_file_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "amx_archive_has_file": (C.c_int, [_P, C.c_char_p]),
    "amx_archive_read_file": (C.c_int, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "amx_archive_write_file": (C.c_int, [_P, C.c_char_p, _P, C.c_size_t, C.c_int]),
    "amx_archive_remove_file": (C.c_int, [_P, C.c_char_p]),
    "amx_feature_cache_write": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, _P, _P, C.c_uint, C.c_int]),
    "amx_feature_cache_read": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_P), C.POINTER(_P)]),
    "amx_feature_cache_write_attributes": (C.c_int, [_P, C.c_char_p, C.c_int, _P, _P, C.c_int]),
    "amx_feature_cache_read_attributes": (C.c_int, [_P, C.c_char_p, C.POINTER(_P)]),
    "amx_device_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "amx_device_free": (None, [_P, _P]),
    "amx_copy_to_device": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "amx_copy_to_host": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "amx_gather_scores": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "amx_stats_accumulate_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "amx_comm_available": (C.c_int, []),
    "amx_comm_unique_id": (C.c_int, [_P]),
    "amx_comm_init": (C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "amx_comm_rank": (C.c_int, [_P]),
    "amx_comm_world": (C.c_int, [_P]),
    "amx_comm_all_reduce_f64_dev": (C.c_int, [_P, _P, C.c_size_t]),
    "amx_comm_destroy": (None, [_P]),
    "amx_counts_to_f64_dev": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "amx_f64_to_counts_dev": (C.c_int, [_P, _P, _P, C.c_size_t]),
}
AMX_COMM_ID_BYTES = 128

_lib = None


def lib():
    """Load librasr_amd.so; raises if it has not been built (there is no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("rasr_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc, gfx950). This package has no CPU fallback." % LIB_PATH)
        # When torch is installed its bundled HIP runtime must be the one the process loads first: a process that loads
        # /opt/rocm's libamdhip64 (through this library) and torch's copy afterwards ends up with two runtimes, and torch's
        # then reports "No HIP GPUs are available".  Tests and bench use torch for device buffers, so import it up front.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != AMX_OK:
        raise AmxError(status, lib().amx_last_error().decode("utf-8", "replace"))
