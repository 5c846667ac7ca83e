"""rasr_amd -- MI355X-native acoustic front-end and emission scorers for RASR.

Python here is plumbing only (ctypes onto librasr_amd.so, torch for device buffers / streams /
torch.distributed); all arithmetic runs in the HIP kernels behind include/amx.h.  The C++
adapters a RASR maintainer links are described in INTEGRATION.md; the classes below mirror the
same reference interfaces for tests and benchmarks:

  MfccExtractor        mfcc.flow network (Tools/FeatureExtraction/share/mfcc.flow)
  GmmFeatureScorer     Mm::FeatureScorer over a Mm::MixtureSet (diagonal-maximum / diagonal-sum)
  NnBatchFeatureScorer Nn::BatchFeatureScorer (nn-batch-feature-scorer)
  FileArchive          Core::FileArchive + Flow cache entries (feature caches between jobs; host IO)
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import (AMX_ACT_NONE, AMX_ACT_RELU, AMX_ACT_SIGMOID, AMX_ACT_TANH, AMX_GMM_BATCH_FLOAT, AMX_GMM_BAUM_WELCH, AMX_GMM_MAX, AMX_GMM_SUM, AMX_GMM_VITERBI,  # noqa: F401
                   AMX_PREC_BF16, AMX_PREC_FP32, AmxError, MfccCfg)

__all__ = ["Context", "MfccExtractor", "GmmFeatureScorer", "NnBatchFeatureScorer", "FileArchive", "AmxError", "read_pms", "write_pms",
           "read_nn_matrix", "write_nn_matrix", "layer_from_parameters", "prior_from_mixture_set", "gmm_estimate",
           "AMX_GMM_VITERBI", "AMX_GMM_BAUM_WELCH"]


def _ptr(a):
    """address of a numpy array or of a (device) torch tensor"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()


def _is_bytes(a):
    """a best-density matrix in its byte form (torch.uint8 tensor)"""
    return a is not None and not isinstance(a, np.ndarray) and a.element_size() == 1


def version():
    """amx_version(): library version, compiler, flags and the hash of the sources it was built from"""
    return _lib.lib().amx_version().decode()


class Comm:
    """The per-epoch exchange between data-parallel ranks (amx_comm_*): RCCL all-reduce of ONE flat f64 device buffer.

    rank 0 creates the 128-byte id (Comm.unique_id()) and hands it to the others -- bench.py broadcasts it through the
    torch.distributed group that also carries its barrier; a RASR trainer would use a file or its own launcher."""

    @staticmethod
    def available():
        return bool(_lib.lib().amx_comm_available())

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * _lib.AMX_COMM_ID_BYTES)()
        _lib.check(_lib.lib().amx_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, ctx, rank, world, unique_id):
        if len(unique_id) != _lib.AMX_COMM_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % _lib.AMX_COMM_ID_BYTES)
        self.L, self.ctx = _lib.lib(), ctx
        buf = (C.c_ubyte * _lib.AMX_COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _lib.check(self.L.amx_comm_init(ctx.h, int(rank), int(world), buf, C.byref(h)))
        self.h = h
        self.rank, self.world = int(rank), int(world)
        ctx._comms.append(self)   # a communicator belongs to its context: Context.close() closes it first (amx_comm_destroy reads the context)

    def _on_torch_stream(self):
        # the tensors handed in were produced on torch's current stream: launch there (the context's own stream is not ordered against it)
        self.ctx.use_torch_stream()

    def all_reduce_f64(self, flat):
        """in-place sum over the ranks of a contiguous float64 device tensor, on torch's current stream"""
        import torch
        if flat.dtype != torch.float64 or not flat.is_contiguous() or not flat.is_cuda:
            raise ValueError("all_reduce_f64 wants a contiguous float64 device tensor")
        self._on_torch_stream()
        _lib.check(self.L.amx_comm_all_reduce_f64_dev(self.h, flat.data_ptr(), flat.numel()))
        return flat

    def counts_to_f64(self, counts, out):
        self._on_torch_stream()
        _lib.check(self.L.amx_counts_to_f64_dev(self.ctx.h, counts.data_ptr(), out.data_ptr(), counts.numel()))

    def f64_to_counts(self, src, counts):
        self._on_torch_stream()
        _lib.check(self.L.amx_f64_to_counts_dev(self.ctx.h, src.data_ptr(), counts.data_ptr(), counts.numel()))

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):   # the context is gone (finalisers run in any order at shutdown): nothing left to destroy safely
                self.L.amx_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One per process and GPU (amx_ctx)."""

    def __init__(self, device=0):
        self.L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self.L.amx_init(device, C.byref(h)))
        self.h = h
        self.device = device
        self._comms = []

    def close(self):
        if getattr(self, "h", None):
            for c in list(getattr(self, "_comms", [])):
                c.close()
            self._comms = []
            self.L.amx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    CONTRACTS = {"off": 0, "fma": 1}

    def set_contract(self, contract):
        """which build of the reference the context's f32 arithmetic follows: "off" (-DMARCH=x86-64) | "fma" (RASR's default build);
        amx_set_contract.  Read by the *_dev entry points at call time and by the front-end / GMM handles at creation."""
        _lib.check(self.L.amx_set_contract(self.h, self.CONTRACTS[contract] if isinstance(contract, str) else int(contract)))
        return self

    def contract(self):
        return {0: "off", 1: "fma"}[int(self.L.amx_get_contract(self.h))]

    def use_torch_stream(self):
        """Launch on torch's current HIP stream (so torch events / allocator ordering apply)."""
        import torch
        handle = torch.cuda.current_stream(self.device).cuda_stream
        # torch's default stream is the legacy NULL stream (handle 0); amx_set_stream(NULL) would select the context's own
        # non-blocking stream, which is NOT ordered against it -> name the legacy stream explicitly (hipStreamLegacy = 1)
        _lib.check(self.L.amx_set_stream(self.h, C.c_void_p(handle if handle else 1)))

    def synchronize(self):
        _lib.check(self.L.amx_synchronize(self.h))

    def gather_scores(self, scores_dev, ld, rows, cols, n_rows=None):
        """scores_dev[rows[i] * ld + cols[i]] for host index arrays -> host float32 array (device gather + one small copy: what a
        decoder's ContextScorer::scores(list) costs against a resident score block); n_rows: rows of the block (default: the
        tensor's first dimension) -- pairs outside [n_rows x ld] are refused"""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        out = np.empty(len(rows), np.float32)
        if n_rows is None:
            n_rows = int(scores_dev.shape[0]) if scores_dev.dim() > 1 else int(scores_dev.numel() // int(ld))
        _lib.check(self.L.amx_gather_scores(self.h, scores_dev.data_ptr(), int(n_rows), int(ld), len(rows), rows.ctypes.data,
                                            cols.ctypes.data, out.ctypes.data))
        return out

    def profile(self, enable=True):
        _lib.check(self.L.amx_profile_enable(self.h, 1 if enable else 0))

    def profile_reset(self):
        _lib.check(self.L.amx_profile_reset(self.h))

    def device_clocks(self, out_dev):
        """enqueue a sample of (s_memtime, s_memrealtime) into out_dev (2 x int64 / uint64 on the device)"""
        _lib.check(self.L.amx_device_clocks_dev(self.h, _ptr(out_dev)))

    def device_clocks_xcd(self, out_dev):
        """the same per XCD: out_dev [8 x 2] int64 / uint64 on the device, zeroed by the caller"""
        _lib.check(self.L.amx_device_clocks_xcd_dev(self.h, _ptr(out_dev)))

    def profile_get(self, kernel):
        ms, n = C.c_double(), C.c_long()
        _lib.check(self.L.amx_profile_get(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def context_window(self, plan, feats, dim, left, right, out, out_stride):
        _lib.check(self.L.amx_context_window_dev(self.h, plan.h, _ptr(feats), dim, left, right, _ptr(out), out_stride))

    # ---- feature back-end: device matrices [total_frames x ld], segmented like the plan
    def normalize(self, plan, feats, in_ld, dim, out, out_ld, variance=False, length=0, right=0):
        """signal-normalization: type mean / mean-and-variance; length = 0 is the whole segment"""
        _lib.check(self.L.amx_normalize_dev(self.h, plan.h, _ptr(feats), in_ld, dim,
                                            _lib.AMX_NORM_MEAN_AND_VARIANCE if variance else _lib.AMX_NORM_MEAN, length, right, _ptr(out), out_ld))

    def normalize_ex(self, plan, feats, in_ld, dim, out, out_ld, type, level=0, length=0, right=0):
        """signal-normalization types 2 divide-by-mean, 3 level (component `level`), 4 mean-and-variance-1D"""
        _lib.check(self.L.amx_normalize_ex_dev(self.h, plan.h, _ptr(feats), in_ld, dim, type, level, length, right, _ptr(out), out_ld))

    VECTOR_NORMALIZATIONS = {"amplitude-spectrum-energy": 0, "energy": 1, "maximum": 2, "mean-energy": 3, "mean": 4, "variance": 5}

    def vector_normalize(self, kind, feats, in_ld, n, dim, out, out_ld):
        """signal-vector-f32-<kind>-normalization on [n x dim] device views (in place on the identical view is allowed)"""
        _lib.check(self.L.amx_vector_normalize_dev(self.h, self.VECTOR_NORMALIZATIONS[kind], _ptr(feats), in_ld, n, dim, _ptr(out), out_ld))

    VECTOR_FUNCTIONS = {"log": 0, "log-plus": 1, "ln": 2, "exp": 3, "power": 4, "sqrt": 5, "cos": 6, "addition": 7, "multiplication": 8,
                        "quantize": 9, "abs": 10, "minimum": 11, "maximum": 12}

    def vector_function(self, kind, parameter, feats, in_ld, n, dim, out, out_ld):
        """generic-vector-f32-<kind> on [n, dim] device views (row strides in_ld / out_ld)"""
        _lib.check(self.L.amx_vector_function_dev(self.h, self.VECTOR_FUNCTIONS[kind], float(parameter), _ptr(feats), in_ld, n, dim, _ptr(out), out_ld))

    def regression(self, plan, feats, in_ld, dim, out, out_ld, order=1, right=2):
        """signal-delay (copy margin) + signal-regression of the given order over 2 * right + 1 frames"""
        _lib.check(self.L.amx_regression_dev(self.h, plan.h, _ptr(feats), in_ld, dim, order, right, _ptr(out), out_ld))

    def matrix_multiply(self, matrix, rows, cols, feats, in_ld, T, out, out_ld):
        """signal-matrix-multiplication-f32: out[t] = M feats[t]"""
        _lib.check(self.L.amx_matrix_multiply_dev(self.h, _ptr(matrix), rows, cols, _ptr(feats), in_ld, T, _ptr(out), out_ld))

    def stats_accumulate(self, scores, T, M, best_state, counts, score_sum):
        _lib.check(self.L.amx_stats_accumulate_dev(self.h, _ptr(scores), T, M, _ptr(best_state), _ptr(counts), _ptr(score_sum)))


class _Plan:
    def __init__(self, owner, sample_offsets):
        self.owner = owner
        self.L = owner.L
        off = np.ascontiguousarray(sample_offsets, dtype=np.int64)
        self.n_seg = len(off) - 1
        h = C.c_void_p()
        _lib.check(self.L.amx_mfcc_plan_create(owner.h, self.n_seg, off.ctypes.data, C.byref(h)))
        self.h = h
        self.total_frames = int(self.L.amx_mfcc_plan_total_frames(h))
        fo = np.zeros(self.n_seg + 1, np.int64)
        _lib.check(self.L.amx_mfcc_plan_frame_offsets(h, fo.ctypes.data))
        self.frame_offsets = fo

    def __del__(self):
        try:
            if self.h:
                self.L.amx_mfcc_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass


class GammatoneExtractor:
    """signal-gammatone -> signal-temporalintegration [-> signal-spectralintegration -> generic-vector-f32-power ->
    signal-cosine-transform].  Keyword names are the fields of amx_gammatone_cfg (= the nodes' parameters)."""

    def __init__(self, ctx, **kw):
        self.ctx, self.L = ctx, (ctx.L if ctx is not None else _lib.lib())
        cfg = _lib.GammatoneCfg()
        self.L.amx_gammatone_default_cfg(C.byref(cfg))
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown gammatone parameter %r" % k)
            setattr(cfg, k, _tuning(v) if k == "tuning" else v)
        self.cfg = cfg
        h = C.c_void_p()
        _lib.check(self.L.amx_gammatone_create(ctx.h if ctx is not None else None, C.byref(cfg), C.byref(h)))
        self.h = h
        info = _lib.GammatoneInfo()
        _lib.check(self.L.amx_gammatone_describe(h, C.byref(info)))
        self.info, self.n_out, self.channels = info, info.n_out, info.channels

    def __del__(self):
        try:
            if self.h:
                self.L.amx_gammatone_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def n_frames(self, n_samples):
        return int(self.L.amx_gammatone_n_frames(self.h, n_samples))

    def tables(self):
        cf, co = np.zeros(self.channels, np.float32), np.zeros((self.channels, 4), np.float32)
        _lib.check(self.L.amx_gammatone_tables(self.h, cf.ctypes.data, co.ctypes.data))
        return cf, co

    def run(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        out = np.zeros((self.n_frames(len(pcm)), self.n_out), np.float32)
        _lib.check(self.L.amx_gammatone_run(self.h, pcm.ctypes.data, len(pcm), out.ctypes.data))
        return out

    def run_batch_dev(self, sample_offsets, pcm_dev, out_dev, filtered_dev=None):
        """torch tensors on the device; sample_offsets: host int64 [n_seg + 1]"""
        off = np.ascontiguousarray(sample_offsets, dtype=np.int64)
        _lib.check(self.L.amx_gammatone_run_batch_dev(self.h, len(off) - 1, off.ctypes.data, pcm_dev.data_ptr(), out_dev.data_ptr(),
                                                      filtered_dev.data_ptr() if filtered_dev is not None else None))


class MfccExtractor:
    """The mfcc.flow chain.  Keyword names follow the Flow node parameters."""

    def __init__(self, ctx, nr_cepstrum_coefficients=16, filter_width=268.258, sample_rate=16000.0, alpha=1.0,
                 length=0.025, shift=0.01, maximum_input_size=0.025, apply_scale=True, spacing=0.0,
                 warp_differential_unit=True, normalize=False, front_end="mfcc", nr_autocorrelation_coefficients=0,
                 intensity_loudness_power=0.33, type="triangular", boundary="stretch-to-cover", warping_function="mel", tuning=None):
        """front_end "mfcc" (mfcc.flow), "mfplp" (mfplp.flow: pass normalize=True and nr_autocorrelation_coefficients) or "plp"
        (plp.flow: MfccExtractor.plp() fills in that file's values); type / boundary / warping_function are signal-filterbank's"""
        self.ctx, self.L = ctx, ctx.L
        cfg = MfccCfg(sample_rate, length, shift, alpha, maximum_input_size, int(apply_scale), filter_width, spacing,
                      int(warp_differential_unit), nr_cepstrum_coefficients, int(normalize),
                      {"mfcc": 0, "mfplp": 1, "plp": 2}[front_end], int(nr_autocorrelation_coefficients), float(intensity_loudness_power),
                      {"triangular": 0, "trapeze": 1}[type], {"stretch-to-cover": 0, "include-boundary": 1, "emphasize-boundary": 2}[boundary],
                      {"mel": 0, "bark": 1}[warping_function], _tuning(tuning))
        h = C.c_void_p()
        _lib.check(self.L.amx_mfcc_create(ctx.h, C.byref(cfg), C.byref(h)))
        self.h = h
        info = _lib.MfccInfo()
        _lib.check(self.L.amx_mfcc_describe(h, C.byref(info)))
        self.info = info
        self.n_ceps, self.n_filters = info.n_ceps, info.n_filters
        self.frame_len, self.frame_shift, self.fft_len = info.frame_len, info.frame_shift, info.fft_len

    def __del__(self):
        try:
            if self.h:
                self.L.amx_mfcc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @classmethod
    def plp(cls, ctx, nr_cepstrum_coefficients=13, nr_autocorrelation_coefficients=13, sample_rate=16000.0, spacing=0.93853,
            filter_width=3.8, **kw):
        """plp.flow: 20 ms Hamming window, no preemphasis, trapeze / include-boundary / bark filter bank, equal loudness"""
        return cls(ctx, nr_cepstrum_coefficients=nr_cepstrum_coefficients, nr_autocorrelation_coefficients=nr_autocorrelation_coefficients,
                   sample_rate=sample_rate, spacing=spacing, filter_width=filter_width, alpha=0.0, length=0.02, maximum_input_size=0.02,
                   normalize=True, front_end="plp", type="trapeze", boundary="include-boundary", warping_function="bark", **kw)

    def equal_loudness(self):
        out = np.zeros(self.info.n_transform_inputs, np.float64)
        _lib.check(self.L.amx_mfcc_equal_loudness(self.h, out.ctypes.data))
        return out

    def n_frames(self, n_samples):
        return int(self.L.amx_mfcc_n_frames(self.h, n_samples))

    def frame_start_time(self, frame):
        return float(self.L.amx_mfcc_frame_start_time(self.h, frame))

    def tables(self):
        i = self.info
        win = np.zeros(i.frame_len, np.float32)
        fs, fe, fo = np.zeros(i.n_filters, np.int32), np.zeros(i.n_filters, np.int32), np.zeros(i.n_filters + 1, np.int32)
        _lib.check(self.L.amx_mfcc_tables(self.h, None, None, None, fo.ctypes.data, None, None))
        fw = np.zeros(int(fo[-1]), np.float32)
        dct = np.zeros((i.n_transform, i.n_transform_inputs), np.float32)
        _lib.check(self.L.amx_mfcc_tables(self.h, win.ctypes.data, fs.ctypes.data, fe.ctypes.data, fo.ctypes.data,
                                          fw.ctypes.data, dct.ctypes.data))
        return dict(window=win, filter_start=fs, filter_end=fe, filter_offset=fo, filter_weights=fw, dct=dct)

    def run(self, pcm):
        """host path: one segment of samples -> [n_frames, n_ceps].  An int16 array goes through the s16 entry point (the samples as
        the audio file holds them, widened inside the kernel); anything else is taken as f32 sample values"""
        s16 = isinstance(pcm, np.ndarray) and pcm.dtype == np.int16
        pcm = np.ascontiguousarray(pcm, dtype=np.int16 if s16 else np.float32)
        out = np.zeros((self.n_frames(len(pcm)), self.n_ceps), np.float32)
        fn = self.L.amx_mfcc_run_s16 if s16 else self.L.amx_mfcc_run
        _lib.check(fn(self.h, pcm.ctypes.data, len(pcm), out.ctypes.data))
        return out

    def run_batch(self, pcms):
        s16 = all(isinstance(p, np.ndarray) and p.dtype == np.int16 for p in pcms) and len(pcms) > 0
        pcms = [np.ascontiguousarray(p, dtype=np.int16 if s16 else np.float32) for p in pcms]
        outs = [np.zeros((self.n_frames(len(p)), self.n_ceps), np.float32) for p in pcms]
        n = len(pcms)
        ip = (C.c_void_p * n)(*[p.ctypes.data for p in pcms])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        ln = np.array([len(p) for p in pcms], np.int64)
        fn = self.L.amx_mfcc_run_batch_s16 if s16 else self.L.amx_mfcc_run_batch
        _lib.check(fn(self.h, n, C.cast(ip, C.c_void_p), ln.ctypes.data, C.cast(op, C.c_void_p)))
        return outs

    def plan(self, sample_offsets):
        return _Plan(self, sample_offsets)

    def run_plan(self, plan, pcm_dev, ceps_dev):
        """device path: concatenated PCM tensor (float32 or int16) -> [total_frames, n_ceps] tensor (both resident in HBM)"""
        import torch
        if pcm_dev.dtype == torch.int16:
            _lib.check(self.L.amx_mfcc_run_plan_dev_s16(self.h, plan.h, _ptr(pcm_dev), _ptr(ceps_dev)))
        else:
            _lib.check(self.L.amx_mfcc_run_plan_dev(self.h, plan.h, _ptr(pcm_dev), _ptr(ceps_dev)))


def _tuning(t):
    """tuning=None | "key=value,..." | dict -> bytes for the `tuning` field of the ABI structs (A/B runs and tests)"""
    if not t:
        return None
    if isinstance(t, dict):
        t = ",".join("%s=%s" % (k, v) for k, v in t.items())
    return t.encode()


def _gmm_struct(model, mixture_weight_scale, gaussian_scale, keep, tuning=None):
    m = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in model.items()}
    assert m["mix_offsets"].dtype == np.uint32 and m["dens_index"].dtype == np.uint32
    assert m["dens_mean"].dtype == np.uint32 and m["dens_cov"].dtype == np.uint32
    assert m["log_weight"].dtype == np.float64 and m["means"].dtype == np.float32 and m["variances"].dtype == np.float32
    keep.append(m)
    return _lib.GmmModel(int(m["dim"]), len(m["mix_offsets"]) - 1, len(m["dens_mean"]), m["means"].shape[0],
                         m["variances"].shape[0], m["mix_offsets"].ctypes.data, m["dens_index"].ctypes.data,
                         m["log_weight"].ctypes.data, m["dens_mean"].ctypes.data, m["dens_cov"].ctypes.data,
                         m["means"].ctypes.data, m["variances"].ctypes.data, mixture_weight_scale, gaussian_scale, _tuning(tuning))


class GmmFeatureScorer:
    """Mm::FeatureScorer over a mixture set; feature_scorer_type in {"diagonal-maximum", "diagonal-sum",
    "batch-diagonal-maximum-float", "SIMD-diagonal-maximum"}.

    model: dict(dim, mix_offsets u32[M+1], dens_index u32[sumK], log_weight f64[sumK], dens_mean u32[D],
    dens_cov u32[D], means f32[n_mean,dim], variances f32[n_cov,dim]).
    """

    def __init__(self, ctx, model, feature_scorer_type="diagonal-maximum", mixture_weight_scale=1.0, gaussian_scale=1.0, tuning=None):
        # ctx = None: host-only handle (prepared tables, accumulator files); scoring then fails with AMX_ERR_STATE
        self.ctx, self.L = ctx, (ctx.L if ctx is not None else _lib.lib())
        self.mode = {"diagonal-maximum": AMX_GMM_MAX, "diagonal-sum": AMX_GMM_SUM,
                     "batch-diagonal-maximum-float": AMX_GMM_BATCH_FLOAT, "SIMD-diagonal-maximum": _lib.AMX_GMM_SIMD, "batch-diagonal-maximum-int": _lib.AMX_GMM_BATCH_INT,
                     "batch-diagonal-maximum-fast": _lib.AMX_GMM_BATCH_INT, "preselection-batch-float": _lib.AMX_GMM_PRESELECTION_FLOAT,
                     "preselection-batch-int": _lib.AMX_GMM_PRESELECTION_INT}[feature_scorer_type]
        keep = []
        st = _gmm_struct(model, mixture_weight_scale, gaussian_scale, keep, tuning)   # tuning: amx_gmm_model.tuning, e.g. "screen=0"
        h = C.c_void_p()
        _lib.check(self.L.amx_gmm_create(ctx.h if ctx is not None else None, C.byref(st), C.byref(h)))
        self.h = h
        self.n_mix, self.dim = st.n_mix, st.dim
        self._nk, self._ncov = int(keep[0]["mix_offsets"][-1]), st.n_cov

    def __del__(self):
        try:
            if self.h:
                self.L.amx_gmm_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def nMixtures(self):
        return int(self.L.amx_gmm_n_mixtures(self.h))

    def dimension(self):
        return int(self.L.amx_gmm_dimension(self.h))

    def tables(self):
        a, b, c = np.zeros(self._nk, np.float32), np.zeros((self._ncov, self.dim), np.float32), np.zeros(self._ncov, np.float32)
        _lib.check(self.L.amx_gmm_tables(self.h, a.ctypes.data, b.ctypes.data, c.ctypes.data))
        return a, b, c

    def score(self, feats, want_best=True):
        """host path: feats [T, dim] -> (scores [T, M], best_density [T, M])"""
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        T = feats.shape[0]
        sc = np.zeros((T, self.n_mix), np.float32)
        best = np.zeros((T, self.n_mix), np.uint32) if want_best else None
        _lib.check(self.L.amx_gmm_score(self.h, self.mode, feats.ctypes.data, T, sc.ctypes.data, _ptr(best)))
        return (sc, best) if want_best else sc

    def score_dev(self, feats_dev, T, scores_dev, best_dev=None):
        _lib.check(self.L.amx_gmm_score_dev(self.h, self.mode, _ptr(feats_dev), T, _ptr(scores_dev), _ptr(best_dev)))

    def score_stats_dev(self, feats_dev, T, scores_dev, best_density, best_state, counts, score_sum):
        """diagonal-maximum scores plus best state / per-state counts / sum of best scores (arg-min fused where possible); a
        one-byte best_density tensor (torch.uint8) selects the byte form of the matrix (amx_gmm_score_stats_u8_dev)"""
        fn = self.L.amx_gmm_score_stats_u8_dev if _is_bytes(best_density) else self.L.amx_gmm_score_stats_dev
        _lib.check(fn(self.h, _ptr(feats_dev), T, _ptr(scores_dev), _ptr(best_density), _ptr(best_state), _ptr(counts), _ptr(score_sum)))

    def simd_scaling(self):
        """quantisation scaling factor of the SIMD-diagonal-maximum scorer"""
        return float(self.L.amx_gmm_simd_scaling(self.h))

    def set_preselection(self, clusters=256, select_clusters=32, iterations=5, backoff_score=40000.0):
        """density-clustering parameters of preselection-batch-float (Mm/DensityClustering.cc:21-35)"""
        _lib.check(self.L.amx_gmm_set_preselection(self.h, clusters, select_clusters, iterations, backoff_score))

    def preselection_clustering(self):
        """(cluster index of every mixture entry uint32[sum K], cluster means float32[n_clusters, dim])"""
        fn = self.L.amx_gmm_preselection_int_clustering if self.mode == _lib.AMX_GMM_PRESELECTION_INT else self.L.amx_gmm_preselection_clustering
        n = C.c_int(0)
        _lib.check(fn(self.h, C.byref(n), None, None))
        cof = np.zeros(self._nk, np.uint32)
        cm = np.zeros((n.value, self.dim), np.float32)
        _lib.check(fn(self.h, C.byref(n), cof.ctypes.data, cm.ctypes.data))
        return cof, cm

    def screen_counts(self, enable=True):
        """(densities evaluated exactly, (frame, mixture) pairs) of the fused screened scorer since the last call; sets counting on / off"""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        _lib.check(self.L.amx_gmm_screen_counts(self.h, 1 if enable else 0, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def accumulator_size(self):
        return int(self.L.amx_gmm_accumulator_size(self.h))

    def write_accumulator(self, acc, path):
        """flat f64 accumulator (numpy, host) -> binary MIXSET estimator file (Mm::MixtureSetEstimator::write)"""
        a = np.ascontiguousarray(acc, dtype=np.float64)
        if a.size != self.accumulator_size():
            raise ValueError("accumulator has %d entries, expected %d" % (a.size, self.accumulator_size()))
        _lib.check(self.L.amx_gmm_accumulator_write(self.h, a.ctypes.data, os.fsencode(path)))

    def read_accumulator(self, path):
        a = np.zeros(self.accumulator_size(), np.float64)
        _lib.check(self.L.amx_gmm_accumulator_read(self.h, os.fsencode(path), a.ctypes.data))
        return a

    def best_density_dev(self, feats_dev, T, mixture_dev, best_density_dev, scores_dev=None):
        """AssigningContextScorer::bestDensity(e) for one mixture per frame: best_density_dev[t] (u32) and, optionally, scores_dev[t]"""
        _lib.check(self.L.amx_gmm_best_density_dev(self.h, _ptr(feats_dev), T, _ptr(mixture_dev), _ptr(best_density_dev), _ptr(scores_dev)))

    def accumulate_dev(self, feats_dev, T, mixture_dev, best_density_dev, best_density_ld, acc_dev):
        """Viterbi statistics (weights, sum x, sum x^2 in f64) into the flat accumulator acc_dev; best_density_dev u32 or bytes"""
        fn = self.L.amx_gmm_accumulate_u8_dev if _is_bytes(best_density_dev) else self.L.amx_gmm_accumulate_dev
        _lib.check(fn(self.h, _ptr(feats_dev), T, _ptr(mixture_dev), _ptr(best_density_dev), best_density_ld, _ptr(acc_dev)))


    def accumulate_weighted_dev(self, mode, feats_dev, T, mixture_dev, weight_dev, best_density_dev, best_density_ld, acc_dev):
        """weighted Viterbi (mode AMX_GMM_VITERBI) or Baum-Welch (AMX_GMM_BAUM_WELCH) statistics; weight_dev f64 per frame or None"""
        _lib.check(self.L.amx_gmm_accumulate_weighted_dev(self.h, mode, _ptr(feats_dev), T, _ptr(mixture_dev), _ptr(weight_dev),
                                                          _ptr(best_density_dev), best_density_ld, _ptr(acc_dev)))


class NnBatchFeatureScorer:
    """Nn::BatchFeatureScorer: Ws[l] is [out, in] (RASR weights_[0] is the same memory, [in x out] col-major)."""

    def __init__(self, ctx, Ws, biases, activations, log_prior=None, priori_scale=1.0, precision="bf16", class_to_output=None, tuning=None):
        """class_to_output: Nn::ClassLabelWrapper mapping [n_classes] (emission -> network output, -1 = disregarded class)"""
        self.ctx, self.L = ctx, ctx.L
        n = len(Ws)
        self._Ws = [np.ascontiguousarray(w, dtype=np.float32) for w in Ws]
        self._bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
        self._ind = np.array([w.shape[1] for w in self._Ws], np.int32)
        self._outd = np.array([w.shape[0] for w in self._Ws], np.int32)
        self._act = np.array(activations, np.int32)
        self._lp = None if log_prior is None else np.ascontiguousarray(log_prior, dtype=np.float32)
        self._map = None if class_to_output is None else np.ascontiguousarray(class_to_output, dtype=np.int32)
        Wp = (C.c_void_p * n)(*[w.ctypes.data for w in self._Ws])
        Bp = (C.c_void_p * n)(*[b.ctypes.data for b in self._bs])
        st = _lib.FfnnModel(n, self._ind.ctypes.data, self._outd.ctypes.data, C.cast(Wp, C.c_void_p), C.cast(Bp, C.c_void_p),
                            self._act.ctypes.data, _ptr(self._lp), priori_scale,
                            {"fp32": AMX_PREC_FP32, "bf16": AMX_PREC_BF16, "bf16x3": _lib.AMX_PREC_BF16X3, "f16mx": _lib.AMX_PREC_F16MX}[precision],
                            0 if self._map is None else len(self._map), _ptr(self._map), _tuning(tuning))   # e.g. tuning="tile=3"
        h = C.c_void_p()
        _lib.check(self.L.amx_ffnn_create(ctx.h, C.byref(st), C.byref(h)))
        self.h = h
        self.in_dim, self.out_dim = int(self._ind[0]), int(self.L.amx_ffnn_output_dim(h))
        self.hidden_dim = int(self.L.amx_ffnn_hidden_dim(h))

    def __del__(self):
        try:
            if self.h:
                self.L.amx_ffnn_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def nMixtures(self):
        return int(self.L.amx_ffnn_output_dim(self.h))

    def score(self, feats):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        T = feats.shape[0]
        out = np.zeros((T, self.out_dim), np.float32)
        _lib.check(self.L.amx_ffnn_score(self.h, feats.ctypes.data, T, out.ctypes.data))
        return out

    def forward_hidden_dev(self, feats_dev, feats_stride, T, act_dev):
        """Nn::OnDemandFeatureScorer::forwardHiddenLayers for T frames: act_dev [T, hidden_dim] f32"""
        _lib.check(self.L.amx_ffnn_forward_hidden_dev(self.h, _ptr(feats_dev), feats_stride, T, _ptr(act_dev)))

    def forward_dev(self, feats_dev, feats_stride, T, out_dev, top="softmax"):
        """Nn::NeuralNetworkForwardNode: the top layer's output [T, out_last]; top "linear" (W x + b - alpha log prior) or "softmax"""
        _lib.check(self.L.amx_ffnn_forward_dev(self.h, _ptr(feats_dev), feats_stride, T, _ptr(out_dev), {"linear": 0, "softmax": 1}[top]))

    def score_on_demand_dev(self, act_dev, n_pairs, frame_dev, emission_dev, scores_dev):
        """output layer for (frame, emission) pairs only: scores_dev [n_pairs]"""
        _lib.check(self.L.amx_ffnn_score_on_demand_dev(self.h, _ptr(act_dev), n_pairs, _ptr(frame_dev), _ptr(emission_dev), _ptr(scores_dev)))

    def effective_precision(self):
        """(precision the handle computes in, block-maximum statistic of its weights): "f16mx" requested on heavy-tailed weights runs "bf16x3"""
        r = C.c_double(0.0)
        p = self.L.amx_ffnn_precision(self.h, C.byref(r))
        return {0: "fp32", 1: "bf16", 2: "bf16x3", 3: "f16mx"}[p], float(r.value)

    def wait_dev(self):
        """amx_ffnn_wait_dev: waits for the handle's stream; raises if a pass of an f16mx scorer left the f16 range"""
        _lib.check(self.L.amx_ffnn_wait_dev(self.h))

    def score_dev(self, feats_dev, feats_stride, T, scores_dev):
        _lib.check(self.L.amx_ffnn_score_dev(self.h, _ptr(feats_dev), feats_stride, T, _ptr(scores_dev)))

    def score_stats_dev(self, feats_dev, feats_stride, T, scores_dev, best_state, counts, score_sum):
        """scores plus best state / per-state counts / sum of best scores (arg-min fused into the output layer)"""
        _lib.check(self.L.amx_ffnn_score_stats_dev(self.h, _ptr(feats_dev), feats_stride, T, _ptr(scores_dev), _ptr(best_state),
                                                   _ptr(counts), _ptr(score_sum)))


def dc_detection(pcm, sample_rate=16000.0, min_dc_length=0.0125, max_dc_increment=0.9, min_non_dc_segment_length=0.02, maximal_output_size=4096,
                 merge=False):
    """signal-dc-detection over one segment (host): [(first sample, length)] of the blocks the node lets through; merge=True joins blocks
    without a gap (the ranges to frame separately)"""
    x = np.ascontiguousarray(pcm, dtype=np.float32)
    L = _lib.lib()
    n = C.c_longlong(0)
    args = (x.ctypes.data if len(x) else None, len(x), float(sample_rate), float(min_dc_length), float(max_dc_increment),
            float(min_non_dc_segment_length), int(maximal_output_size), int(merge))
    _lib.check(L.amx_dc_detection(*args, None, None, 0, C.byref(n)))
    st, ln = np.zeros(n.value, np.int64), np.zeros(n.value, np.int64)
    _lib.check(L.amx_dc_detection(*args, st.ctypes.data, ln.ctypes.data, n.value, C.byref(n)))
    return list(zip(st.tolist(), ln.tolist()))


def class_labels_init(n_classes, disregard=()):
    """Nn::ClassLabelWrapper::initMapping: (mapping int32[n_classes], number of classes to accumulate)"""
    dis = np.ascontiguousarray(list(disregard), dtype=np.int32)
    mapping = np.zeros(n_classes, np.int32)
    nt = C.c_int(0)
    _lib.check(_lib.lib().amx_class_labels_init(n_classes, _ptr(dis) if len(dis) else None, len(dis), mapping.ctypes.data, C.byref(nt)))
    return mapping, int(nt.value)


def _read_vector(fn, path, ctype, dtype):
    n, p = C.c_int(0), C.c_void_p()
    L = _lib.lib()
    _lib.check(getattr(L, fn)(os.fsencode(path), C.byref(n), C.byref(p)))
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), shape=(max(n.value, 1),))[:n.value].astype(dtype).copy()
    finally:
        L.amx_free(p)


def read_prior(path):
    """Nn::Prior::read: Math::Vector<f32> file (xml, or bin:<path>) -> log-prior float32[n]"""
    return _read_vector("amx_nn_vector_read_f32", path, C.c_float, np.float32)


def write_prior(path, log_prior):
    v = np.ascontiguousarray(log_prior, dtype=np.float32)
    _lib.check(_lib.lib().amx_nn_vector_write_f32(os.fsencode(path), len(v), v.ctypes.data))


def read_class_labels(path):
    """Nn::ClassLabelWrapper::load: Math::Vector<s32> xml file -> mapping int32[n_classes]"""
    return _read_vector("amx_nn_vector_read_s32", path, C.c_int, np.int32)


def write_class_labels(path, mapping):
    v = np.ascontiguousarray(mapping, dtype=np.int32)
    _lib.check(_lib.lib().amx_nn_vector_write_s32(os.fsencode(path), len(v), v.ctypes.data))


def precomputed_score_dev(ctx, feats_dev, feats_stride, T, n_classes, class_to_output_dev, log_prior_dev, prior_scale, scores_dev):
    """Nn::PrecomputedFeatureScorer: -x[out(e)] + alpha * logPrior[out(e)]"""
    _lib.check(ctx.L.amx_precomputed_score_dev(ctx.h, _ptr(feats_dev), feats_stride, T, n_classes, _ptr(class_to_output_dev),
                                               _ptr(log_prior_dev), prior_scale, _ptr(scores_dev)))


def _mixture_set_to_dict(h):
    L = _lib.lib()
    try:
        v = _lib.GmmModel()
        _lib.check(L.amx_mixture_set_view(h, C.byref(v)))
        nk_arr = np.ctypeslib.as_array(C.cast(v.mix_offsets, C.POINTER(C.c_uint32)), shape=(v.n_mix + 1,)).copy()
        nk = int(nk_arr[-1])

        def arr(p, t, shape):
            n = int(np.prod(shape))
            if n == 0:
                return np.zeros(shape, np.dtype(t))
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(t)), shape=shape).copy()

        return dict(dim=v.dim, mix_offsets=nk_arr, dens_index=arr(v.dens_index, C.c_uint32, (nk,)),
                    log_weight=arr(v.log_weight, C.c_double, (nk,)), dens_mean=arr(v.dens_mean, C.c_uint32, (v.n_dens,)),
                    dens_cov=arr(v.dens_cov, C.c_uint32, (v.n_dens,)), means=arr(v.means, C.c_float, (v.n_mean, v.dim)),
                    variances=arr(v.variances, C.c_float, (v.n_cov, v.dim)))
    finally:
        L.amx_mixture_set_destroy(h)


def read_pms(path):
    """text mixture set -> model dict (see GmmFeatureScorer)"""
    h = C.c_void_p()
    _lib.check(_lib.lib().amx_pms_read(path.encode(), C.byref(h)))
    return _mixture_set_to_dict(h)


def gmm_estimate(model, acc, **cfg):
    """Re-estimation (and, with split=True, splitting) from the flat f64 statistics: Mm::AbstractMixtureSetEstimator::estimate /
    Mm::MixtureSetSplitter::split.  `model` is the model dict the statistics were accumulated with; keyword arguments are the
    fields of amx_gmm_estimate_cfg.  Returns the new model dict."""
    L = _lib.lib()
    c = _lib.GmmEstimateCfg()
    L.amx_gmm_estimate_cfg_default(C.byref(c))
    for k, v in cfg.items():
        if not hasattr(c, k):
            raise TypeError("unknown estimate option %r" % k)
        setattr(c, k, v)
    keep = []
    st = _gmm_struct(model, 1.0, 1.0, keep)
    a = np.ascontiguousarray(acc, dtype=np.float64)
    nk = int(np.asarray(model["mix_offsets"])[-1])
    need = nk + st.n_mean * (1 + st.dim) + st.n_cov * (1 + st.dim)
    if a.size != need:
        raise ValueError("accumulator has %d entries, expected %d" % (a.size, need))
    h = C.c_void_p()
    _lib.check(L.amx_gmm_estimate(C.byref(st), a.ctypes.data, C.byref(c), C.byref(h)))
    return _mixture_set_to_dict(h)


def write_pms(model, path):
    keep = []
    st = _gmm_struct(model, 1.0, 1.0, keep)
    _lib.check(_lib.lib().amx_pms_write(C.byref(st), path.encode()))



class FileArchive:
    """Core::FileArchive ("SP_ARC1") over the C ABI: a RASR feature cache file.  ``mode`` "r" or "w" (read-write,
    created when missing).  Mirrors the calls of Flow::Cache / Core::Archive (src/Flow/Cache.cc, src/Core/Archive.hh)."""

    def __init__(self, path, mode="r"):
        self._h = C.c_void_p()
        m = {"r": _lib.AMX_ARCHIVE_READ, "w": _lib.AMX_ARCHIVE_WRITE}[mode]
        _lib.check(_lib.lib().amx_archive_open(os.fsencode(path), m, C.byref(self._h)))

    def close(self):
        if self._h:
            h, self._h = self._h, C.c_void_p()
            _lib.check(_lib.lib().amx_archive_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def files(self):
        """[(name, size, compressed_size)] in archive order"""
        L, out = _lib.lib(), []
        for i in range(L.amx_archive_n_files(self._h)):
            name, size, comp = C.c_char_p(), C.c_uint32(), C.c_uint32()
            _lib.check(L.amx_archive_file_info(self._h, i, C.byref(name), C.byref(size), C.byref(comp)))
            out.append((name.value.decode(), size.value, comp.value))
        return out

    def __contains__(self, name):
        return bool(_lib.lib().amx_archive_has_file(self._h, name.encode()))

    def read_file(self, name):
        L, p, n = _lib.lib(), C.c_void_p(), C.c_size_t()
        _lib.check(L.amx_archive_read_file(self._h, name.encode(), C.byref(p), C.byref(n)))
        try:
            return C.string_at(p, n.value)
        finally:
            L.amx_free(p)

    def write_file(self, name, data, compress=False):
        data = bytes(data)
        _lib.check(_lib.lib().amx_archive_write_file(self._h, name.encode(), data, len(data), int(compress)))

    def remove_file(self, name):
        _lib.check(_lib.lib().amx_archive_remove_file(self._h, name.encode()))

    # ---- Flow cache entries (vector-f32 packets)
    def write_features(self, segment, feats, times, gather=0xFFFFFFFF, compress=False, attributes=None):
        """feats [n, dim] f32, times [n, 2] f64 (start, end); attributes: optional {name: value} -> '<segment>.attribs'"""
        x = np.ascontiguousarray(feats, dtype=np.float32)
        t = np.ascontiguousarray(times, dtype=np.float64)
        if x.ndim != 2 or t.shape != (x.shape[0], 2):
            raise ValueError("write_features: feats [n, dim] and times [n, 2] expected")
        L = _lib.lib()
        if attributes is not None:
            names = (C.c_char_p * len(attributes))(*[k.encode() for k in attributes])
            vals = (C.c_char_p * len(attributes))(*[str(v).encode() for v in attributes.values()])
            _lib.check(L.amx_feature_cache_write_attributes(self._h, segment.encode(), len(attributes), names, vals, int(compress)))
        _lib.check(L.amx_feature_cache_write(self._h, segment.encode(), x.shape[0], x.shape[1], x.ctypes.data, t.ctypes.data,
                                             int(gather), int(compress)))

    def read_features(self, segment):
        """-> (feats [n, dim] f32, times [n, 2] f64)"""
        L = _lib.lib()
        n, d, px, pt = C.c_int(), C.c_int(), C.c_void_p(), C.c_void_p()
        _lib.check(L.amx_feature_cache_read(self._h, segment.encode(), C.byref(n), C.byref(d), C.byref(px), C.byref(pt)))
        try:
            cnt = n.value * d.value
            x = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_float)), shape=(max(cnt, 1),))[:cnt].copy()
            t = np.ctypeslib.as_array(C.cast(pt, C.POINTER(C.c_double)), shape=(max(2 * n.value, 1),))[:2 * n.value].copy()
            return x.reshape(n.value, d.value), t.reshape(n.value, 2)
        finally:
            L.amx_free(px)
            L.amx_free(pt)

    def read_attributes(self, segment):
        """'<segment>.attribs' -> {name: value}"""
        import xml.etree.ElementTree as ET
        L, p = _lib.lib(), C.c_void_p()
        _lib.check(L.amx_feature_cache_read_attributes(self._h, segment.encode(), C.byref(p)))
        try:
            root = ET.fromstring(C.string_at(p).decode())
        finally:
            L.amx_free(p)
        return {e.get("name"): e.get("value") for e in root.iter("flow-attribute")}


def read_nn_matrix(path):
    """binary Math::Matrix<f32> (RASR NN layer parameter file) -> numpy [rows, cols]"""
    L = _lib.lib()
    r, c, p = C.c_int(), C.c_int(), C.c_void_p()
    _lib.check(L.amx_nn_matrix_read(path.encode(), C.byref(r), C.byref(c), C.byref(p)))
    try:
        n = r.value * c.value
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(max(n, 1),))[:n].copy()
        return a.reshape(r.value, c.value)
    finally:
        L.amx_free(p)


def write_nn_matrix(path, m):
    m = np.ascontiguousarray(m, dtype=np.float32)
    _lib.check(_lib.lib().amx_nn_matrix_write(path.encode(), m.shape[0], m.shape[1], m.ctypes.data))


def layer_from_parameters(params, has_bias=True):
    """parameter matrix [out, has_bias + in] (column 0 = bias) -> (W [out, in], bias [out])"""
    p = np.ascontiguousarray(params, dtype=np.float32)
    out, cols = p.shape
    W = np.zeros((out, cols - int(has_bias)), np.float32)
    b = np.zeros(out, np.float32)
    _lib.check(_lib.lib().amx_nn_layer_from_parameters(p.ctypes.data, out, cols, int(has_bias), W.ctypes.data, b.ctypes.data))
    return W, b


def prior_from_mixture_set(model):
    """Nn::Prior::setFromMixtureSet: log prior per mixture from the mixture weights"""
    keep = []
    st = _gmm_struct(model, 1.0, 1.0, keep)
    out = np.zeros(st.n_mix, np.float32)
    _lib.check(_lib.lib().amx_prior_from_mixture_set(C.byref(st), out.ctypes.data))
    return out
