/*
 * amx.h -- C ABI of librasr_amd.so: MI355X (gfx950) acoustic front-end and emission scorers
 * for RASR.  Plain C, no RASR / torch types.  This is the drop-in boundary: the RASR-side
 * adapters subclass Flow::Node and Mm::FeatureScorer and forward to these
 * entry points; nothing else of RASR is replaced.
 *
 * Reference interfaces replaced (paths relative to /root/reference/src):
 *   amx_mfcc_*   the Flow sub-network Tools/FeatureExtraction/share/mfcc.flow:8-34, i.e. the nodes
 *                Signal/Preemphasis.cc:108-124, Signal/SlidingAlgorithmNode.hh:60-79 (signal-window),
 *                Signal/FastFourierTransform.hh:311-320, Signal/ComplexVectorFunction.hh:234-243,
 *                Signal/Filterbank.cc:860-876, Flow/SimpleFunction.hh:497-507,
 *                Signal/CosineTransform.cc:213-228 -- driven per segment by
 *                Speech/DataExtractor.cc:101-111 (FeatureExtractor::processSegment).
 *   amx_gmm_*    Mm::FeatureScorer::getScorer(x)->score(e) as implemented by
 *                Mm/GaussDiagonalMaximumFeatureScorer.cc:116-141 (max) and :252-298 (log-add),
 *                constructed by Mm/FeatureScorerFactory.hh:72-82 from a Mm::MixtureSet.
 *   amx_ffnn_*   Nn::BatchFeatureScorer (Nn/BatchFeatureScorer.cc:45-171): forward of
 *                Nn::NeuralNetwork<f32> (Nn/NeuralNetwork.cc:313-331) with the softmax disabled and
 *                the scaled log-prior removed from the output bias; score = -activation.
 *   amx_stats_*  the per-partition accumulator files + `combine-mixture-set-estimators`
 *                (Tools/AcousticModelTrainer/AcousticModelTrainer.cc:317-325) -- here a flat device
 *                buffer the caller all-reduces (RCCL) once per epoch.
 *
 * Conventions
 *   - every function returns an amx_status (0 = ok, < 0 = error); amx_last_error() gives the
 *     thread-local message of the last failure.  Nothing exits or throws across the boundary.
 *   - "host" pointers are ordinary process memory (pinned recommended); "dev" pointers are HBM
 *     addresses on the context's device (hipMalloc / torch tensor data_ptr).
 *   - all matrices handed over the boundary are row-major [frames x dim] f32.
 *   - scores are negative log-likelihoods (Mm::Score = f32, smaller is better).
 *   - a context is bound to one device and one HIP stream; calls on one context must be
 *     externally serialised (RASR's corpus loop is single threaded, Speech/Recognizer.cc:271-281).
 */
#ifndef RASR_AMD_AMX_H
#define RASR_AMD_AMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    AMX_OK              = 0,
    AMX_ERR_INVALID     = -1, /* bad argument / configuration (reference would criticalError) */
    AMX_ERR_UNSUPPORTED = -2, /* valid RASR configuration this build has no kernel for */
    AMX_ERR_DEVICE      = -3, /* HIP runtime failure, no gfx950 device, out of memory */
    AMX_ERR_STATE       = -4  /* call sequence violates the protocol (reference would require()) */
} amx_status;

typedef struct amx_ctx  amx_ctx;
typedef struct amx_mfcc amx_mfcc;
typedef struct amx_mfcc_plan amx_mfcc_plan;
typedef struct amx_gmm  amx_gmm;
typedef struct amx_ffnn amx_ffnn;

/* ------------------------------------------------------------------ context */

/* "rasr_amd <version> (gfx950; <compiler>; <flags>; src=<12 hex digits>)": the source hash is the SHA-1 of the files of rasr_amd/csrc/ and
 * include/amx.h the library was built from (rasr_amd/csrc/Makefile), so that a measurement can be tied to a build. */
const char* amx_version(void);
const char* amx_last_error(void);
/* Creates a context on HIP device `device_ordinal` with its own non-blocking stream. */
int  amx_init(int device_ordinal, amx_ctx** out);
void amx_destroy(amx_ctx* ctx);
/* Run all subsequent launches of this context on the caller's hipStream_t (e.g. torch's current
 * stream); NULL restores the context's own stream.  To run on the legacy default stream pass hipStreamLegacy
 * ((hipStream_t)1): the handle 0 that frameworks report for it would read as NULL here. */
int amx_set_stream(amx_ctx* ctx, void* hip_stream);
int amx_synchronize(amx_ctx* ctx);

/* WHICH BUILD OF THE REFERENCE this context's f32 arithmetic follows (round 6; cmake_resources/CompileOptions.cmake:21-48).  RASR has
 * two arithmetics, decided by how it was compiled:
 *   AMX_CONTRACT_OFF  -DMARCH=x86-64 (or a host / compiler without fused multiply-adds): every product and every sum rounds once;
 *   AMX_CONTRACT_FMA  its DEFAULT configuration, -march=native on an FMA host with GCC (-ffp-contract=fast; clang's default `on` fuses
 *                     within a statement: treat clang builds as FMA too): `sum += a * b` is ONE fused multiply-add at the sites
 *                     oracle/orc.h marks ORC_FMAF / ORC_FMA -- the GMM distance (Mm/GaussDiagonalMaximumFeatureScorer.cc:144-218,
 *                     Mm/BatchFeatureScorer.cc:164-650), Signal::Regression (Signal/Regression.cc:24-65), Math::Vector's dot product
 *                     (Math/Vector.hh:94-101: signal-matrix-multiplication-f32, the cosine transform), Filter::apply
 *                     (Signal/Filterbank.cc:27-50), preemphasis with alpha != 1, the filter bank's geometry, the AR-to-cepstrum
 *                     recursion, the gammatone design / cascade / integrations, the amplitude-spectrum-energy normalisation.
 * The setting is read (a) by every *_dev entry point that takes the context directly (amx_regression_dev, amx_matrix_multiply_dev,
 * amx_vector_normalize_dev, ...) at call time, (b) by amx_mfcc_create / amx_gammatone_create / amx_gmm_create at creation (a handle
 * keeps the arithmetic it was created with; amx_gmm_model.tuning "contract=off|fma" overrides it per model).  Default: OFF.
 * An adapter compiled INSIDE RASR's build knows the answer at compile time: pass AMX_CONTRACT_OF_THIS_BUILD right after amx_init --
 * the macro is evaluated in the translation unit that includes this header, i.e. with RASR's own flags.
 * Bit-exact against oracle/liboracle.so (OFF) / oracle/liboracle_fma.so (FMA) wherever the OFF arithmetic is (tests/test_contract_gpu.py). */
enum { AMX_CONTRACT_OFF = 0, AMX_CONTRACT_FMA = 1 };
#if defined(__FMA__) && (defined(__GNUC__) || defined(__clang__)) && !defined(AMX_ADAPTER_NO_FP_CONTRACT)
#define AMX_CONTRACT_OF_THIS_BUILD AMX_CONTRACT_FMA /* -mfma / -march=native on an FMA host, GCC or clang defaults (define AMX_ADAPTER_NO_FP_CONTRACT when RASR is built with -ffp-contract=off) */
#else
#define AMX_CONTRACT_OF_THIS_BUILD AMX_CONTRACT_OFF
#endif
int amx_set_contract(amx_ctx* ctx, int contract);
int amx_get_contract(const amx_ctx* ctx); /* AMX_CONTRACT_OFF | AMX_CONTRACT_FMA; < 0: ctx is NULL */
/* "contract=off: RASR built with -DMARCH=x86-64 ..." -- one line an adapter logs next to the reference's own "Scaling factor" lines */
const char* amx_contract_description(int contract);
/* Per-kernel timing with HIP events on the context's stream (used by bench.py for the roofline
 * line).  enable=1 makes every launch of the named hot kernels record start/stop events. */
int amx_profile_enable(amx_ctx* ctx, int enable);
int amx_profile_reset(amx_ctx* ctx);
/* kernel: "mfcc", "gmm", "gmm_dist", "gmm_combine", "ffnn_gemm" (all layers), "ffnn_gemm_max"
 * (largest layer only).  Synchronises the stream.  avg_ms = mean launch duration. */
int amx_profile_get(amx_ctx* ctx, const char* kernel, double* avg_ms, long* n_launches);

/* ------------------------------------------------------------------ MFCC front-end */

typedef struct {
    double sample_rate;            /* Hz */
    double win_len_s;              /* signal-window length        (mfcc.flow: 0.025) */
    double win_shift_s;            /* signal-window shift         (mfcc.flow: 0.01)  */
    double preemph_alpha;          /* signal-preemphasis alpha    (mfcc.flow: 1.00)  */
    double fft_max_input_s;        /* maximum-input-size          (mfcc.flow: 0.025) */
    int    apply_scale;            /* apply-scale, default 1: spectrum * 1/(f32)fs   */
    double mel_filter_width;       /* filter-width, default 268.258 (mel)            */
    double mel_spacing;            /* spacing, default 0 => 0.5 * width              */
    int    warp_differential_unit; /* warp-differential-unit, default 1              */
    int    n_ceps;                 /* signal-cosine-transform nr-outputs             */
    int    dct_normalize;          /* normalize, default 0                           */
    /* front_end AMX_FRONT_END_MFPLP = mfplp.flow (Tools/FeatureExtraction/share/mfplp.flow:9-47): amplitude ->
     * generic-vector-f32-power 2 -> mel filter bank -> generic-vector-f32-power plp_power -> signal-cosine-transform
     * (input-type N-plus-one, nr-outputs n_autocorrelation, normalize) -> signal-autocorrelation-to-autoregression
     * (Math/LevinsonLse.cc:35-70) -> signal-autoregression-to-cepstrum (Signal/AutoregressionToCepstrum.cc:21-35,
     * nr-outputs n_ceps).  A frame whose Levinson recursion meets a zero prediction error (digital silence) is an error in
     * the reference ("Failed to calculate the autoregression coefficients."); it comes out as NaNs here. */
    int    front_end;              /* AMX_FRONT_END_MFCC (0, default), AMX_FRONT_END_MFPLP or AMX_FRONT_END_PLP  */
    int    n_autocorrelation;      /* nr-autocorrelation-coefficients = LPC order + 1 (MF-PLP, PLP)                */
    double plp_power;              /* intensity-loudness-law value, default 0.33 (MF-PLP, PLP)                     */
    /* The other parameters of signal-filterbank (Signal/Filterbank.cc:700-745), usable with every front end; the defaults (0) are
     * mfcc.flow's.  warp-center-positions stays true (the reference's default; stretch-to-cover accepts nothing else).
     * front_end AMX_FRONT_END_PLP = plp.flow (Tools/FeatureExtraction/share/plp.flow): Hamming 20 ms, no preemphasis node
     * (preemph_alpha 0), power spectrum -> trapeze / include-boundary / bark filter bank (filter-width 3.8, spacing 0.93853 = 20
     * filters at 16 kHz; 0.973442 = 15 filters at 8 kHz) -> the vector extended by copies of its first and last element
     * (generic-vector-f32-split / -concat) -> equal-loudness preemphasis (signal-vector-f32-continuous-transform,
     * Signal/VectorTransform.cc:36-83, f = equal-loudness(bark^-1(index / sample-rate)), operation multiplies) -> ^plp_power ->
     * the MF-PLP tail (cosine transform N-plus-one, Levinson, LPC cepstrum). */
    int    filter_type;            /* type: AMX_FILTER_TRIANGULAR (0) or AMX_FILTER_TRAPEZE                        */
    int    boundary;               /* boundary: AMX_BOUNDARY_STRETCH_TO_COVER (0), _INCLUDE, _EMPHASIZE            */
    int    warping;                /* warping-function: AMX_WARP_MEL (0) or AMX_WARP_BARK                          */
    /* Kernel selection for A/B runs and tests, "key=value,key=value" (NULL: defaults; an unknown key fails amx_mfcc_create).  None
     * of them changes a result beyond the parity bars.  fft=stockham|mfma|r16 (LDS radix-4 butterflies | the 256-point transform as
     * two matrix products | radix-16 register butterflies, four frames per wave: mfcc.flow with a 512-point transform only),
     * prefetch=1|0 (a wave fetches its next frame's samples while it transforms the current one; default 1), wgs=N (workgroups
     * per CU), lpc=regs|lds (LPC-cepstrum recursion in registers | the LDS kernel). */
    const char* tuning;
} amx_mfcc_cfg;
enum { AMX_FRONT_END_MFCC = 0, AMX_FRONT_END_MFPLP = 1, AMX_FRONT_END_PLP = 2 };
enum { AMX_FILTER_TRIANGULAR = 0, AMX_FILTER_TRAPEZE = 1 };
enum { AMX_BOUNDARY_STRETCH_TO_COVER = 0, AMX_BOUNDARY_INCLUDE = 1, AMX_BOUNDARY_EMPHASIZE = 2 };
enum { AMX_WARP_MEL = 0, AMX_WARP_BARK = 1 };

typedef struct {
    int    frame_len, frame_shift, fft_len, n_bins, n_filters, n_ceps;
    double fft_output_sample_rate; /* attribute "sample-rate" after the FFT node = N/fs */
    double mel_max;                /* warped maximum frequency */
    int    n_transform;            /* rows of the cosine-transform table: n_ceps (MFCC) or n_autocorrelation (MF-PLP, PLP) */
    int    n_transform_inputs;     /* its columns: n_filters, or n_filters + 2 (PLP: first and last filter output duplicated) */
} amx_mfcc_info;

void amx_mfcc_default_cfg(amx_mfcc_cfg* cfg); /* the values of mfcc.flow + node defaults, 16 ceps */
void amx_mfplp_default_cfg(amx_mfcc_cfg* cfg); /* the values of mfplp.flow, 13 autocorrelation / 13 cepstrum coefficients */
void amx_plp_default_cfg(amx_mfcc_cfg* cfg);   /* the values of plp.flow at 16 kHz, 13 autocorrelation / 13 cepstrum coefficients */
int  amx_mfcc_create(amx_ctx* ctx, const amx_mfcc_cfg* cfg, amx_mfcc** out);
void amx_mfcc_destroy(amx_mfcc* h);
int  amx_mfcc_describe(const amx_mfcc* h, amx_mfcc_info* info);
/* frames produced for a segment of n_samples (framing + flush rule of Signal/WindowBuffer.cc:84-125) */
long amx_mfcc_n_frames(const amx_mfcc* h, long n_samples);
/* start time of frame k relative to the segment start, accumulated exactly like
 * WindowBuffer::get (bufferStartTime_ += shift / sampleRate) */
double amx_mfcc_frame_start_time(const amx_mfcc* h, long frame);
/* host copies of the tables the kernel uses (any pointer may be NULL):
 * window[frame_len], filter_start/end[n_filters], filter_offset[n_filters+1],
 * filter_weights[filter_offset[n_filters]], dct[n_transform*n_transform_inputs] */
int amx_mfcc_tables(const amx_mfcc* h, float* window, int* filter_start, int* filter_end,
                    int* filter_offset, float* filter_weights, float* dct);
/* PLP: the equal-loudness factors [n_transform_inputs] (f64, as the reference's f(i)); AMX_ERR_STATE for the other front ends */
int amx_mfcc_equal_loudness(const amx_mfcc* h, double* factors);

/* One segment, host buffers: pcm f32 (s16 sample values, unscaled, Flow/TypeConverter.hh:35-43)
 * -> ceps [n_frames x n_ceps].  Includes H2D/D2H. */
int amx_mfcc_run(amx_mfcc* h, const float* pcm_host, long n_samples, float* ceps_host);
/* Batch of segments, host buffers. */
int amx_mfcc_run_batch(amx_mfcc* h, int n_seg, const float* const* pcm_host, const long* n_samples,
                       float* const* ceps_host);
/* The same with the samples as the audio file holds them: s16, widened to f32 without scaling inside the kernel exactly like the
 * converter node in front of the chain (Flow/TypeConverter.hh:35-43) -- 320 instead of 640 bytes per frame over the host link and
 * out of HBM; results are bit-identical to the f32 entry points on the same sample values. */
int amx_mfcc_run_s16(amx_mfcc* h, const int16_t* pcm_host, long n_samples, float* ceps_host);
int amx_mfcc_run_batch_s16(amx_mfcc* h, int n_seg, const int16_t* const* pcm_host, const long* n_samples,
                           float* const* ceps_host);

/* Device-resident batches.  A plan fixes the segmentation: segment u occupies samples
 * [sample_offsets[u], sample_offsets[u+1]) of one concatenated PCM buffer and frames
 * [frame_offsets[u], frame_offsets[u+1]) of one [total_frames x n_ceps] output. */
int  amx_mfcc_plan_create(amx_mfcc* h, int n_seg, const long* sample_offsets /*[n_seg+1]*/, amx_mfcc_plan** out);
void amx_mfcc_plan_destroy(amx_mfcc_plan* p);
long amx_mfcc_plan_total_frames(const amx_mfcc_plan* p);
int  amx_mfcc_plan_frame_offsets(const amx_mfcc_plan* p, long* frame_offsets /*[n_seg+1]*/);
int  amx_mfcc_run_plan_dev(amx_mfcc* h, const amx_mfcc_plan* p, const float* pcm_dev, float* ceps_dev);
int  amx_mfcc_run_plan_dev_s16(amx_mfcc* h, const amx_mfcc_plan* p, const int16_t* pcm_dev, float* ceps_dev);

/* ------------------------------------------------------------------ sample stream in front of the feature chains (samples.flow)
 * signal-dc-detection (Signal::DcDetection, src/Signal/DcDetection.cc:90-235): a host-side scan over one segment's samples that tells
 * the caller which sample ranges reach the feature chain.  Runs of at least min-dc-length seconds whose samples stay within
 * max-dc-increment of the last accepted sample are dropped, and so are non-DC stretches shorter than min-non-dc-segment-length (node
 * defaults .0125 / 0.9 / .02, maximal-output-size 4096; samples.flow uses .0125 / 0.9 / .026).  max_dc_increment = 0 disables the
 * detection (every sample is accepted).  Output: the node's blocks as (first sample, length) in stream order -- merge = 0: exactly
 * the vectors the node emits; merge = 1: neighbouring blocks without a time gap joined, i.e. the ranges the window buffer behind it
 * frames without an intermediate flush (one entry of an amx_mfcc_plan's sample_offsets each).  starts / lengths may be NULL to count. */
int amx_dc_detection(const float* pcm, long long n_samples, double sample_rate, double min_dc_length_s, float max_dc_increment,
                     double min_non_dc_segment_length_s, int maximal_output_size, int merge, long long* starts, long long* lengths,
                     long long capacity, long long* n_blocks);

/* ------------------------------------------------------------------ gammatone front-end
 * The three nodes of Signal/Module.cc:169-173 -- signal-gammatone (Signal/GammaTone.cc:20-231), signal-temporalintegration
 * (Signal/TemporalIntegration.cc:60-84 over Signal/TimeWindowBuffer.cc:52-125) and signal-spectralintegration
 * (Signal/SpectralIntegration.cc:55-74) -- as one front end, PCM in, one vector per 10 ms frame out, with the usual tail
 * (generic-vector-f32-power, signal-cosine-transform) optional.  Field names are the nodes' parameter names; the defaults are the
 * nodes' own.  Temporal integration frames follow the window node's rule (short last frames, amx_gammatone_n_frames); the state
 * of the filter cascade starts at zero with every segment (the node resets at end of segment). */
typedef struct {
    double sample_rate;      /* Hz */
    int    cascade;          /* signal-gammatone cascade, default 4 (0..8)                                   */
    double min_freq;         /* minfreq, default 100                                                         */
    double max_freq;         /* maxfreq, default 6000                                                        */
    double q;                /* q, default 9.264491981582191 (the 0 Hz bandwidth l = 24.7 is not a parameter) */
    int    channels;         /* channels, default 50                                                         */
    int    cf_mode;          /* cfmode: AMX_GAMMATONE_HUMAN (0, default) or AMX_GAMMATONE_ERB                */
    double warp_freq_break;  /* warp-freqbreak, default 6600                                                 */
    double warping_factor;   /* warping-factor, default 1 (the upper end of the warping is sample_rate / 2)  */
    int    ti_window;        /* signal-temporalintegration type: AMX_WINDOW_HANNING (0) or AMX_WINDOW_RECTANGULAR */
    double ti_length_s;      /* length (seconds)                                                             */
    double ti_shift_s;       /* shift (seconds)                                                              */
    int    si_window;        /* signal-spectralintegration type                                              */
    int    si_length;        /* length in channels; 0 = node absent                                          */
    int    si_shift;         /* shift in channels                                                            */
    double power;            /* generic-vector-f32-power value (e.g. 0.1 for the 10th root); 0 = node absent */
    int    n_ceps;           /* signal-cosine-transform nr-outputs; 0 = node absent                          */
    int    dct_normalize;    /* normalize                                                                    */
    /* "contract=off|fma" (NULL: the context's arithmetic, amx_set_contract; OFF for a host-only handle): which build of the reference
     * the design (centre frequencies, warping), the filter cascade, both integrations and the cosine transform follow -- bit for bit
     * in either mode (oracle/orc_gammatone.c marks the contracted sites).  Unknown keys / values fail amx_gammatone_create. */
    const char* tuning;
} amx_gammatone_cfg;
enum { AMX_GAMMATONE_HUMAN = 0, AMX_GAMMATONE_ERB = 1 };
enum { AMX_WINDOW_HANNING = 0, AMX_WINDOW_RECTANGULAR = 1 };
typedef struct {
    int channels, cascade, frame_len, frame_shift;
    int si_channels; /* channels after spectral integration */
    int n_out;       /* output dimension */
} amx_gammatone_info;
typedef struct amx_gammatone amx_gammatone;
void amx_gammatone_default_cfg(amx_gammatone_cfg* cfg);
int  amx_gammatone_create(amx_ctx* ctx, const amx_gammatone_cfg* cfg, amx_gammatone** out); /* ctx NULL: host-only (tables, geometry) */
void amx_gammatone_destroy(amx_gammatone* h);
int  amx_gammatone_describe(const amx_gammatone* h, amx_gammatone_info* info);
long amx_gammatone_n_frames(const amx_gammatone* h, long n_samples);
/* centre frequencies [channels] and filter coefficients [channels][4] = a0, a1, b1, b2 (GammaTone::Coefficient); nullable */
int amx_gammatone_tables(const amx_gammatone* h, float* center_freq, float* coefficients);
/* one segment, host buffers: pcm f32 -> out [n_frames x n_out] */
int amx_gammatone_run(amx_gammatone* h, const float* pcm_host, long n_samples, float* out_host);
/* a batch of segments resident in HBM: segment u = samples [sample_offsets[u], sample_offsets[u+1]) of pcm_dev (offsets on the
 * host), its frames follow those of segment u - 1 in out_dev [total frames x n_out].  filtered_dev (nullable, [total samples x
 * channels]) receives the signal-gammatone node's own output. */
int amx_gammatone_run_batch_dev(amx_gammatone* h, int n_seg, const long* sample_offsets, const float* pcm_dev, float* out_dev,
                                float* filtered_dev);

/* Sliding-window concatenation of feature frames per segment (f1 "next" row:
 * signal-vector-f32-sequence-concatenation, Signal/SlidingWindow.hh:66-76 copy margin policy):
 * out[t] = [x[t-left] .. x[t+right]] with indices clamped to the segment.  out_dev is
 * [total_frames x out_stride] f32 (out_stride >= (left+right+1)*dim, padding zero-filled). */
int amx_context_window_dev(amx_ctx* ctx, const amx_mfcc_plan* p, const float* feats_dev, int dim,
                           int left, int right, float* out_dev, int out_stride);

/* Feature back-end between the front-end and the scorers, on device-resident
 * [total_frames x ld] f32 matrices segmented like the plan.  in/out may point into wider matrices (column offset by
 * pointer arithmetic, row stride *_ld), which is how "generic-vector-f32-concat" of features and derivatives is laid out.
 * Aliasing: the input and output views must not share memory -- disjoint column ranges of one wide matrix are fine, anything
 * else is rejected with AMX_ERR_INVALID; the one exception is whole-segment normalisation (length = 0) in place on the
 * identical view.
 *
 * signal-normalization (src/Signal/Normalization.cc:46-66,120-187): per segment, type mean or mean-and-variance;
 * length = 0: whole segment (length="infinite" right="infinite"); otherwise a sliding window of `length` frames with the
 * output point `right` frames from its newest end (0 <= right < length), statistics kept as the reference's running
 * f64 sums (the last `right` frames of a segment leave with the statistics of the last window, as in the reference). */
#define AMX_NORM_MEAN 0
#define AMX_NORM_MEAN_AND_VARIANCE 1
int amx_normalize_dev(amx_ctx* ctx, const amx_mfcc_plan* plan, const float* in_dev, int in_ld, int dim, int type,
                      int length, int right, float* out_dev, int out_ld);
/* the node's other types on the same window skeleton (src/Signal/Normalization.cc:100-110,196-262; NormalizationNode `type` /
 * `level`): divide-by-mean (x / mean; the reference stops with "One of the mean components is zero." where this yields inf or
 * NaN), level (out[level] = x[level] - max over the window, other components unchanged) and mean-and-variance-1D (one mean and
 * standard deviation over all components of the window).  type mean / mean-and-variance are forwarded to amx_normalize_dev. */
#define AMX_NORM_DIVIDE_BY_MEAN 2
#define AMX_NORM_LEVEL 3
#define AMX_NORM_MEAN_AND_VARIANCE_1D 4
int amx_normalize_ex_dev(amx_ctx* ctx, const amx_mfcc_plan* plan, const float* in_dev, int in_ld, int dim, int type, int level,
                         int length, int right, float* out_dev, int out_ld);
/* signal-vector-f32-{amplitude-spectrum-energy,energy,maximum,mean-energy,mean,variance}-normalization
 * (src/Signal/VectorNormalization.hh:31-163, registered in src/Signal/Module.cc:108-113): every vector on its own -- divided by
 * sqrt of its (Parseval / plain / mean) energy or by its maximum, or shifted to zero mean (and scaled to unit deviation).  A zero
 * statistic gives inf / NaN like the reference's unguarded division.  [n_vectors x dim] views with row strides; in place is
 * allowed on the identical view. */
enum { AMX_VNORM_AMPLITUDE_SPECTRUM_ENERGY = 0, AMX_VNORM_ENERGY = 1, AMX_VNORM_MAXIMUM = 2, AMX_VNORM_MEAN_ENERGY = 3, AMX_VNORM_MEAN = 4,
       AMX_VNORM_VARIANCE = 5 };
int amx_vector_normalize_dev(amx_ctx* ctx, int type, const float* in_dev, int in_ld, long n_vectors, int dim, float* out_dev, int out_ld);
/* generic-vector-f32-<function> (Flow::SimpleFunctionNode over src/Flow/SimpleFunction.hh:40-345; e.g. the x 500 scaling and the
 * quantisation of mfcc.standard_system.flow, the power nodes of mfplp.flow): one parameter, every element on its own.  Addition,
 * multiplication, quantize (rint(v / p) * p; rint(v) for p = 1 or 0), abs, minimum, maximum, sqrt and power (the node's unqualified
 * pow on floats = ::pow(double, double) narrowed) follow the reference's arithmetic exactly; log (log10), log-plus (log10(v + p)), ln,
 * exp and cos are the device's f32 functions, a few ulp from glibc's.  Views and in-place rule as for amx_vector_normalize_dev. */
enum { AMX_VFUNC_LOG = 0, AMX_VFUNC_LOG_PLUS = 1, AMX_VFUNC_LN = 2, AMX_VFUNC_EXP = 3, AMX_VFUNC_POWER = 4, AMX_VFUNC_SQRT = 5, AMX_VFUNC_COS = 6,
       AMX_VFUNC_ADDITION = 7, AMX_VFUNC_MULTIPLICATION = 8, AMX_VFUNC_QUANTIZE = 9, AMX_VFUNC_ABS = 10, AMX_VFUNC_MINIMUM = 11,
       AMX_VFUNC_MAXIMUM = 12 };
int amx_vector_function_dev(amx_ctx* ctx, int kind, float parameter, const float* in_dev, int in_ld, long n_vectors, int dim, float* out_dev,
                            int out_ld);
/* signal-delay (max-size = 2*right+1, margin-policy copy, margin-condition present-not-empty; src/Signal/Delay.hh:33-47)
 * + signal-regression order 1 or 2 (src/Signal/Regression.cc:25-68), as wired in derivationWithRegression.flow. */
int amx_regression_dev(amx_ctx* ctx, const amx_mfcc_plan* plan, const float* in_dev, int in_ld, int dim, int order,
                       int right, float* out_dev, int out_ld);
/* signal-matrix-multiplication-f32 (src/Signal/MatrixMult.hh:246-255; LDA in lda.flow): out[t] = M in[t], M [rows x cols]
 * row-major on the device (read the file with amx_nn_matrix_read: same binary Math::Matrix<f32> format). */
int amx_matrix_multiply_dev(amx_ctx* ctx, const float* matrix_dev, int rows, int cols, const float* in_dev, int in_ld,
                            int T, float* out_dev, int out_ld);

/* ------------------------------------------------------------------ diagonal-covariance GMM */

typedef struct {
    int dim, n_mix, n_dens, n_mean, n_cov;
    const uint32_t* mix_offsets; /* [n_mix+1]  mixture m owns entries [mix_offsets[m], mix_offsets[m+1]) */
    const uint32_t* dens_index;  /* [mix_offsets[n_mix]]  density index (Mm::Mixture::densityIndex) */
    const double*   log_weight;  /* [mix_offsets[n_mix]]  log weights (Mm::Weight = f64) */
    const uint32_t* dens_mean;   /* [n_dens]  Mm::GaussDensity::meanIndex */
    const uint32_t* dens_cov;    /* [n_dens]  Mm::GaussDensity::covarianceIndex */
    const float*    means;       /* [n_mean x dim] */
    const float*    variances;   /* [n_cov  x dim] diagonal variances */
    double          mixture_weight_scale; /* mixture-weight-scale (Core::ParameterFloat = f64; the scorer keeps it as f32), default 1 */
    double          gaussian_scale;       /* gaussian-scale (f64; the scorer keeps (f32)sqrt of the f64 value,
                                           * Mm/GaussDiagonalMaximumFeatureScorer.cc:52), default 1 */
    /* "key=value,key=value" (NULL: defaults).  Keys AND values are checked by amx_gmm_create -- an unknown key, a number that is not one,
     * a value the key does not take fail the creation (AMX_ERR_INVALID): a typo must not silently select another kernel or arithmetic.
     * Read by amx_gmm_create only (amx_pms_write, amx_gmm_estimate, amx_prior_from_mixture_set ignore it).
     *
     * contract=off|fma -- WHICH BUILD OF THE REFERENCE the scorer follows (without the key: the context's, amx_set_contract -- OFF unless
     *   the adapter said otherwise; see there for what the two builds are):
     *   off  RASR configured with -DMARCH=x86-64, or built on a host without FMA units: every f32 operation of the distance
     *        (Mm/GaussDiagonalMaximumFeatureScorer.cc:144-218) rounds once;
     *   fma  RASR's DEFAULT configuration (cmake_resources/CompileOptions.cmake:39-48: -march=native) built with GCC on an FMA host:
     *        `sum += df * df` is ONE fused multiply-add (vfmadd231ps / vfmadd231ss); about a fifth of the d = 40 distances differ in
     *        the last bit from the other build.  One vector operation fewer per dimension in the exact stage: the fused GMM kernel
     *        runs 4.60 -> 4.18 ms on BASELINE config 5's shard (profiles/r05/gmm_contract_ab.log) -- bench.py's default is fma because
     *        it is the reference's default build, and the line names the mode.
     *   Every mode takes both (round 6).  Bit-exact, in the named build: AMX_GMM_MAX scores and density indices, AMX_GMM_BATCH_FLOAT,
     *   AMX_GMM_PRESELECTION_FLOAT (clustering included), amx_gmm_best_density_dev, the quantised scorers (integer arithmetic; their
     *   host-side gaussLogNormFactor follows the contract).  AMX_GMM_SUM: the DISTANCES and the best density follow the contract bit
     *   for bit; the log-add score itself is a streaming sum with the device's expf / logf -- within 1e-5 of the reference's two-pass
     *   form in either mode, not bit-exact.  Baum-Welch statistics: 2e-5 (the same expf).  fused_waves=13 (a lab kernel) has no fma
     *   form and fails the creation with AMX_ERR_UNSUPPORTED.  tests/test_gmm_contract_gpu.py, tests/test_contract_gpu.py.
     *
     * Kernel selection for A/B runs and tests -- every path gives the same scores and density indices bit for bit (within a contract):
     * screen=0 (no MFMA / f32 screen: every density evaluated), fused=0 (two-kernel screen path instead of gmm_fused_kernel),
     * screen_kernel=rows|persist|simple, graph=1 (HIP-graph replay of repeated small passes; off by default since round 6: plain launches
     * measured 3-5 % faster back to back and equal with a synchronisation per pass, profiles/r06/graph_ab.log), tied_prune=0|1 (tied models: dense tile
     * kernel | pruned scorer, default -1 adaptive), chunk=N (frames per internal pass, >= 256), fused_waves=8|12|16|13 (13: the
     * wave-specialised kernel), fr=2|4|8|16 (frames per workgroup of the uniform tied kernel), simd_mfma=0 (SIMD / batch-int scorers
     * without the i8 matrix kernel), dist_list=0 (pruned tied scorer: distances from the density-major kernel instead of the
     * list-order one; N >= 2: N frames per wave of the list-order kernel), near_fused=0 (the frame's near densities from
     * tied_near_kernel instead of the list-order kernel's atomic minima), fused_pack=0 (gmm_fused_kernel reads operand rows packed by
     * gmm_screen_pack_kernel instead of packing its own). */
    const char*     tuning;
} amx_gmm_model;

/* diagonal-maximum / diagonal-sum / batch-diagonal-maximum-float (Mm/BatchFeatureScorer.cc:164-254: pooled
 * covariance only, no best-density output, ignores the two scales like the reference class) /
 * SIMD-diagonal-maximum (Mm::SimdGaussDiagonalMaximumFeatureScorer, Mm/SimdFeatureScorer.cc:68-176 with
 * Mm/IntelOptimization.cc:37-66: means and features times scaling / sigma quantised to u8, integer distance and constant,
 * first minimum, score = 0.5 * min / scaling^2; assigns densities; ignores the two scales like the reference class) /
 * batch-diagonal-maximum-int and -fast (Mm::BatchIntFeatureScorer / BatchUnrolledIntFeatureScorer, Mm/BatchFeatureScorer.cc:375-504:
 * the same u8 quantisation and integer distance, pooled covariance only, constant (s32)(logNorm scale^2 - 2 scale^2 logWeight) formed
 * in f64, score = (f32)min / (2 scale^2) in f32, no best-density output) /
 * preselection-batch-float (see amx_gmm_set_preselection below; pooled covariance only, no best-density output) */
enum { AMX_GMM_MAX = 0, AMX_GMM_SUM = 1, AMX_GMM_BATCH_FLOAT = 2, AMX_GMM_SIMD = 3, AMX_GMM_BATCH_INT = 4, AMX_GMM_PRESELECTION_FLOAT = 5,
       AMX_GMM_PRESELECTION_INT = 6 };

int  amx_gmm_create(amx_ctx* ctx, const amx_gmm_model* model, amx_gmm** out); /* copies everything */
void amx_gmm_destroy(amx_gmm* h);
int  amx_gmm_n_mixtures(const amx_gmm* h);
int  amx_gmm_dimension(const amx_gmm* h);
/* host copies of the prepared scorer tables (Mm/MixtureFeatureScorerElement.cc:21-33,
 * Mm/CovarianceFeatureScorerElement.cc:21-51); any pointer may be NULL */
int amx_gmm_tables(const amx_gmm* h, float* minus2_log_weights, float* inv_sqrt_var, float* log_norm);
/* quantisation scaling factor of the SIMD-diagonal-maximum scorer (SimdGaussDiagonalMaximumFeatureScorer::getScaling,
 * logged as "Scaling factor"); 0 for a host-only handle */
float amx_gmm_simd_scaling(const amx_gmm* h);
/* preselection-batch-float (Mm::BatchPreselectionFloatFeatureScorer, Mm/BatchFeatureScorer.cc:256-318 with
 * Mm::FloatDensityClustering, Mm/DensityClustering.cc / .tcc): batch-diagonal-maximum-float restricted, per frame, to the densities
 * of the `select_clusters` clusters closest to the scaled feature; k-means over the pre-scaled density means (`clusters`
 * clusters, initial clusters drawn with srand(1) / rand(), `iterations` rounds); a mixture without an active density scores
 * `backoff_score`.  Defaults (the reference's): 256 / 32 / 5 / 40000.  The clustering is built on the first preselection call
 * (or amx_gmm_preselection_clustering) and rebuilt after amx_gmm_set_preselection.  amx_gmm_preselection_clustering returns
 * it: cluster_of [sum K_m] (cluster of every mixture entry), cluster_means [n_clusters x dim]; any pointer may be NULL.
 * Parity with the reference binary holds up to DISTANCE TIES between clusters: the reference ranks them with std::sort on the
 * distance alone (selectClusters), whose order among equal distances is implementation-defined; here (and in the oracle) the
 * lower cluster index wins.  Exact ties are rare in f32 and common in the u8 / s32 variant below -- there the selected cluster set,
 * and with it a score or a back-off value, can differ from a given reference build at a tie.  A frame whose distances are all NaN
 * selects the first `select_clusters` clusters (what an index-ordered sort of incomparable keys leaves in front). */
int amx_gmm_set_preselection(amx_gmm* h, int clusters, int select_clusters, int iterations, float backoff_score);
int amx_gmm_preselection_clustering(amx_gmm* h, int* n_clusters, uint32_t* cluster_of, float* cluster_means);
/* AMX_GMM_PRESELECTION_INT = Mm::BatchPreselectionIntFeatureScorer ("preselection-batch-int", Mm/BatchFeatureScorer.cc:514-578):
 * the batch-int scorer over the densities of the select-clusters clusters closest to the QUANTISED feature, with
 * Mm::DensityClustering<u8, s32> over the quantised means (integer distances, cluster means truncated to u8); a mixture without an
 * active density scores (f32)INT_MAX / scale_ -- the int class has no back-off score.  Same parameters
 * (amx_gmm_set_preselection; backoff_score unused).  cluster_means [n_clusters x dim]: the u8 values as floats. */
int amx_gmm_preselection_int_clustering(amx_gmm* h, int* n_clusters, uint32_t* cluster_of, float* cluster_means);
/* Diagnostics of the screened diagonal-maximum scorer (no reference counterpart; bench.py prints survivors per mixture):
 * returns and clears the number of densities evaluated exactly and the number of (frame, mixture) pairs scored since the
 * last call, and switches the counting on (enable != 0) or off for the following calls.  Synchronises the stream.  Zero for
 * models that do not take the fused screened path.  Tied models whose mixtures share one density list (pruned path,
 * gmm_tied.hip): survivors = (density, frame, 64-mixture tile) triples whose weight rows were read, pairs = triples submitted
 * (densities x frames x tiles); always counted. */
int amx_gmm_screen_counts(amx_gmm* h, int enable, unsigned long long* survivors, unsigned long long* pairs);
/* scores [T x n_mix]; best_density (nullable) [T x n_mix] = index within the mixture of the
 * minimising density (AssigningFeatureScorer::ScoreAndBestDensity). */
int amx_gmm_score(amx_gmm* h, int mode, const float* feats_host, int T, float* scores_host, uint32_t* best_density_host);
int amx_gmm_score_dev(amx_gmm* h, int mode, const float* feats_dev, int T, float* scores_dev, uint32_t* best_density_dev);
/* diagonal-maximum scores plus the per-epoch statistics of amx_stats_accumulate_dev for these frames (best state per
 * frame, per-state counts, sum of best scores), the counterpart of amx_ffnn_score_stats_dev.  For screened models the
 * arg-min over the states is taken inside the exact stage, so the [T x n_mix] score matrix is written once and not
 * re-read.  best_density_dev and best_state_dev are nullable. */
int amx_gmm_score_stats_dev(amx_gmm* h, const float* feats_dev, int T, float* scores_dev, uint32_t* best_density_dev,
                            uint32_t* best_state_dev, unsigned long long* state_counts_dev, double* score_sum_dev);
/* The same pass with the best densities as ONE BYTE per (frame, mixture): best_density_dev [T x n_mix] bytes, required, the same
 * index within the mixture (Mm::DensityInMixture, Mm/Types.hh:36, is u32 in RASR; no mixture of a real model comes near 255
 * densities), 0xff where the u32 form writes 0xffffffff.  A quarter of the u32 matrix's MEMORY -- for a 10 000-state model 10 kB
 * instead of 40 kB per frame, next to 40 kB of scores: 0.64 GB instead of 2.56 GB per 63 936-frame pass -- and no gain in TIME:
 * the fused screened kernel writes it directly, but what the best densities cost that kernel (0.3 ms of 4.7) is the vector work of
 * tracking them, not their bytes (tools/gmm_store_ab.py, profiles/r04/gmm_store_ab.log: 4.58-4.62 ms against 4.50-4.76 for u32,
 * 4.22-4.30 without).  Every other path scores into a u32 workspace and narrows it.
 * AMX_ERR_UNSUPPORTED for a model with a mixture of more than 255 densities. */
int amx_gmm_score_stats_u8_dev(amx_gmm* h, const float* feats_dev, int T, float* scores_dev, uint8_t* best_density_dev,
                               uint32_t* best_state_dev, unsigned long long* state_counts_dev, double* score_sum_dev);

/* Viterbi training statistics of Mm::AbstractMixtureSetEstimator::accumulate (Mm/AbstractMixtureSetEstimator.cc:117-125,
 * Mm/GaussDensityEstimator.hh:152-208): frame t, aligned to mixture_dev[t], adds 1 to the weight of its best density
 * (best_density_dev[t * best_density_ld + mixture] as written by amx_gmm_score_dev, or best_density_dev[t] when
 * best_density_ld == 0), x to that density's mean accumulator and x*x to its covariance accumulator, all in f64.
 * acc_dev is ONE flat f64 buffer -- [sum K_m weights][n_mean weights][n_mean x dim sums][n_cov weights][n_cov x dim sums],
 * amx_gmm_accumulator_size() doubles -- so that data-parallel ranks combine their statistics with a single all-reduce
 * (the reference combines per-partition accumulator files offline, Tools/AcousticModelTrainer/AcousticModelTrainer.cc:317-325). */
/* AssigningContextScorer::bestDensity(e) for ONE mixture per frame (Mm/AssigningFeatureScorer.hh:127-131 ->
 * GaussDiagonalMaximumFeatureScorer::calculateScoreAndDensity, Mm/GaussDiagonalMaximumFeatureScorer.cc:116-142): best_density_dev[t] =
 * index within mixture_dev[t] of the density that minimises frame t's score, scores_dev[t] (nullable) that score -- the same bits as
 * entry (t, mixture_dev[t]) of amx_gmm_score_dev's matrices (0xffffffff / Type<f32>::max for a mixture index >= n_mix).  It is what the
 * Viterbi accumulation asks of an assigning scorer: a trainer that scores every state (amx_gmm_score_stats_dev with
 * best_density_dev = NULL: the screened kernel is 0.3 ms of 4.7 faster without the index bookkeeping) needs the density of the
 * ALIGNED state only, and gets it here for amx_gmm_accumulate_dev(..., best_density_ld = 0). */
int amx_gmm_best_density_dev(amx_gmm* h, const float* feats_dev, int T, const uint32_t* mixture_dev, uint32_t* best_density_dev,
                             float* scores_dev);
long amx_gmm_accumulator_size(const amx_gmm* h);
int  amx_gmm_accumulate_dev(amx_gmm* h, const float* feats_dev, int T, const uint32_t* mixture_dev,
                            const uint32_t* best_density_dev, int best_density_ld, double* acc_dev);
/* the same statistics from the byte form of the best-density matrix (amx_gmm_score_stats_u8_dev) */
int  amx_gmm_accumulate_u8_dev(amx_gmm* h, const float* feats_dev, int T, const uint32_t* mixture_dev,
                               const uint8_t* best_density_dev, int best_density_ld, double* acc_dev);

/* The same statistics as a binary "MIXSET" accumulator file, version 2 (Mm::MixtureSetEstimator::write / read,
 * src/Mm/AbstractMixtureSetEstimator.cc:404-508, src/Mm/VectorAccumulator.hh:80-100, src/Mm/MixtureEstimator.cc:140-170):
 * what RASR's accumulate / combine-mixture-set-estimators / estimate actions exchange.  acc_host is the flat buffer of
 * amx_gmm_accumulator_size() doubles (host memory); read requires the file's topology to equal the model's. */
int amx_gmm_accumulator_write(const amx_gmm* h, const double* acc_host, const char* path);
int amx_gmm_accumulator_read(const amx_gmm* h, const char* path, double* acc_host);

/* Weighted Viterbi / Baum-Welch statistics: Mm::AbstractMixtureSetEstimator::accumulate(mixture, x, weight)
 * (Mm/AbstractMixtureSetEstimator.cc:127-147).  weight_dev: per-frame f64 weights (the alignment's posterior weight), NULL = 1.
 *   AMX_GMM_VITERBI     the best density (best_density_dev as above) gets the whole frame weight;
 *   AMX_GMM_BAUM_WELCH  every density k of the aligned mixture gets weight * exp(score(e) - s_k), the density posterior of
 *                       the log-add scorer (Mm/GaussDiagonalMaximumFeatureScorer.cc:291-298), if that product exceeds
 *                       Core::Type<f32>::epsilon; best_density_dev is not used.
 * Sums are weight * x and (weight * x) * x in f64 (plusWeighted / plusSquareWeighted, Mm/Utilities.hh:108-153). */
enum { AMX_GMM_VITERBI = 0, AMX_GMM_BAUM_WELCH = 1 };
int amx_gmm_accumulate_weighted_dev(amx_gmm* h, int mode, const float* feats_dev, int T, const uint32_t* mixture_dev,
                                    const double* weight_dev, const uint32_t* best_density_dev, int best_density_ld,
                                    double* acc_dev);

/* Re-estimation from the (all-reduced) statistics: Mm::AbstractMixtureSetEstimator::estimate
 * (Mm/AbstractMixtureSetEstimator.cc:305-338: densities below the observation-weight limits leave their mixture except
 * the heaviest one, Mm/MixtureEstimator.cc:64-82; remaining densities / means / covariances are renumbered in order of
 * first appearance, :804-817; mixture weights log(w) normalised by logExpNorm, Mm/Mixture.cc:63-74; mean = sum / weight,
 * variance = (sum x^2 - sum_j sum_j^2 / N_j) / N floored at min_variance, Mm/GaussDensityEstimator.cc:148-233), and with
 * cfg->split the acoustic model trainer's splitting step on top of it (Mm::MixtureSetSplitter::split,
 * Mm/MixtureSetSplitter.cc:38-123: means with enough observations become mean +- sqrt(var) * perturbation * f32 epsilon).
 * `topology` is the model the statistics were accumulated with (its means / variances / weights are not read); acc_host the
 * flat buffer of amx_gmm_accumulator_size() doubles.  The result is a mixture set like amx_pms_read's: view it, write it
 * with amx_pms_write, or build the next iteration's scorer from it.  Host work, model-sized, once per epoch. */
typedef struct {
    double min_observation_weight;    /* minimum-observation-weight, default 5 */
    double min_relative_weight;       /* minimum-relative-weight, default 0 */
    double min_variance;              /* minimum-variance, default 0 (compared and stored as f32) */
    int    normalize_mixture_weights; /* normalize-mixture-weights, default 1 */
    int    allow_zero_weights;        /* allow-zero-weights, default 0: a mixture without observations is an error */
    int    split;                     /* 0: estimate only; 1: estimate, then split */
    double split_min_mean_observation_weight;       /* minimum-mean-observation-weight, default 20 */
    double split_min_covariance_observation_weight; /* minimum-covariance-observation-weight, default f32 max (never) */
    double split_perturbation_weight;               /* perturbation-weight, default 0.1 */
    int    split_normalize_mixture_weights;         /* normalize-mixture-weights of the splitter, default 0 */
} amx_gmm_estimate_cfg;
typedef struct amx_mixture_set amx_mixture_set;
void amx_gmm_estimate_cfg_default(amx_gmm_estimate_cfg* cfg);
int  amx_gmm_estimate(const amx_gmm_model* topology, const double* acc_host, const amx_gmm_estimate_cfg* cfg /* NULL: defaults */,
                      amx_mixture_set** out);

/* ------------------------------------------------------------------ mixture-set text files (.pms) */

/* Reader / writer of RASR's text mixture-set format, "#Version: 2.0" (Mm/MixtureSet.cc:141-216,
 * Mm/Mixture.cc:80-105, Mm/MixtureSetTopology.cc:19-30, Mm/GaussDensity.cc:25-70).  Version < 2.0
 * files carry linear weights (converted with log), covariances are stored as (variance, weight)
 * pairs whose product is the diagonal.  The returned object owns its arrays; amx_mixture_set_view
 * fills an amx_gmm_model with pointers into it (scales set to 1). */
int  amx_pms_read(const char* path, amx_mixture_set** out);
int  amx_pms_write(const amx_gmm_model* model, const char* path);
int  amx_mixture_set_view(const amx_mixture_set* ms, amx_gmm_model* view);
void amx_mixture_set_destroy(amx_mixture_set* ms);

/* ------------------------------------------------------------------ feed-forward NN scorer */

enum { AMX_ACT_NONE = 0, AMX_ACT_RELU = 1, AMX_ACT_SIGMOID = 2, AMX_ACT_TANH = 3 };
/* MFMA input type; accumulation is always f32.  AMX_PREC_BF16X3 = split bf16: every operand is hi + lo (two bf16 values) and a
 * product is taken as hi hi + lo hi + hi lo -- three bf16 MFMA products, ~2^-16 relative error per product: the mode that meets
 * the 1e-4 bar of the f32 reference (Math::gemm<f32>, Math/Blas.hh:402-420) at a third of the bf16 rate */
enum { AMX_PREC_FP32 = 0, AMX_PREC_BF16 = 1, AMX_PREC_BF16X3 = 2, AMX_PREC_F16MX = 3 };
/* AMX_PREC_F16MX (round 4): every operand v is h = f16(v) plus an MX-fp6 image of the residual v - h (OCP e2m3, one e8m0 scale per
 * 32 k); a product is h h + fp6(h) fp6(w - h_w) + fp6(v - h_v) fp6(h_w) -- one f16 MFMA product and ONE block-scaled fp6 x fp6 MFMA
 * product for both cross terms (the fp6 image of h is converted from the f16 fragments in registers): 1.5 units of matrix time
 * per product instead of split bf16's 3, worst error 0.03 of the 1e-4 bar on BASELINE config 4 (profiles/r04/emulation_f16_f8.json,
 * tests/test_ffnn_f16mx_gpu.py).
 * Limits: weights and activations must stay inside the f16 range (|v| < 65520, and finite): amx_ffnn_create refuses such weights; a
 * pass that meets such a feature or hidden activation (inf and NaN included) FAILS, its scores are not valid:
 *   amx_ffnn_score (host buffers; it waits for its results) returns AMX_ERR_STATE for that very pass;
 *   the *_dev entry points only enqueue work: call amx_ffnn_wait_dev(h) behind a pass and before its scores are used -- it waits for
 *     the handle's stream and returns AMX_ERR_STATE if the pass (or an earlier one) left the range.  rasr_amd/host/
 *     BatchFeatureScorer.hh does so before a row of the block reaches the decoder (Nn/BatchFeatureScorer.cc:148-171 hands out scores
 *     of a batch it has computed synchronously);
 *   the flag is sticky: every later call on the handle returns AMX_ERR_STATE too (create the scorer with AMX_PREC_BF16X3, which
 *     has no such limit). */

typedef struct {
    int                 n_layers;
    const int*          in_dim;     /* [n_layers] */
    const int*          out_dim;    /* [n_layers] */
    const float* const* W;          /* [n_layers] each [out x in] row-major (== RASR weights_[0], [in x out] col-major) */
    const float* const* bias;       /* [n_layers] each [out] */
    const int*          activation; /* [n_layers] AMX_ACT_*; last layer must be AMX_ACT_NONE */
    const float*        log_prior;  /* nullable [out_last]  (Nn::Prior) */
    float               prior_scale;/* priori-scale alpha */
    int                 precision;  /* AMX_PREC_* */
    /* Nn::ClassLabelWrapper (Nn/ClassLabelWrapper.cc:21-70, used by Nn/BatchFeatureScorer.cc:148-171 and Nn/FeatureScorer.cc):
     * emission (class) e reads network output class_to_output[e]; -1 = disregarded class (`disregard-classes`), which scores
     * Core::Type<f32>::max.  NULL = identity (n_classes is then ignored).  The mapping must be one-to-one and cover every
     * output ("no one-to-one correspondence between network outputs and classes!"); scores are [T x n_classes]. */
    int                 n_classes;
    const int*          class_to_output;
    /* "key=value,key=value" (NULL: defaults); keys and values are checked by amx_ffnn_create like amx_gmm_model.tuning's.
     * Two keys change RESULTS (within the 1e-4 bar of the f32 reference either way):
     *   ksplit=4 (AMX_PREC_F16MX; default 1)  passes of at most 256 frames -- a decoder's ring-buffer fill -- split the K of every
     *        2048-wide layer over four workgroups per tile (two launches: partial sums, then ((P0 + P1) + P2) + P3 and the epilogue):
     *        four times as many CUs pull operands, a 256-frame fill of BASELINE config 4's network takes 0.146 instead of 0.170 ms.
     *        Another association of the sum over k than the default (one accumulator, ascending k): scores differ from the default's
     *        by f32 rounding and are bit-identical among all passes that are split; larger passes of the same handle run the default
     *        order.  Without it, scoring a segment in pieces gives the bits of scoring it whole.
     *   mx_fallback=auto|off (AMX_PREC_F16MX; default auto)  auto: heavy-tailed weights compute in split bf16 (amx_ffnn_precision).
     * Kernel selection for A/B runs and tests -- every tile configuration of a precision gives bit-identical scores: tile=N (GEMM tile
     * configuration: 0 128x128, 2 256x256 with all waves in phase, 8 256x256 with the ping-pong K loop (f16mx; the default for large
     * batches since round 5), 9 256x256 with ONE self-pipelined wave per SIMD (f16mx; measured 6 % behind 8, opt-in), 3 128x64 one tile per
     * CU, 6 128x64 two per CU, 4 / 5 / 7 K-loop variants of 2, 11 / 12 K-loop variants of 3, 14 (f16mx) hidden layers on 64x64 tiles with four
     * K-tiles per barrier -- the default for fills that leave half the CUs without a 128x64 tile; default by layer shape),
     * graph=1 (HIP-graph replay of repeated small passes; off by default since round 6, see amx_gmm_model.tuning), persistent=0, group=TxN (tiles per XCD-aware super-tile), chunk=N (frames per
     * internal pass, >= 256), stagger=N (f16mx output layer of a large batch: XCD x starts x * N * 10 ns late; default 0). */
    const char*         tuning;
} amx_ffnn_model;

int  amx_ffnn_create(amx_ctx* ctx, const amx_ffnn_model* model, amx_ffnn** out);
void amx_ffnn_destroy(amx_ffnn* h);
int  amx_ffnn_input_dim(const amx_ffnn* h);
int  amx_ffnn_output_dim(const amx_ffnn* h);
/* feats [T x in0]; scores [T x out_last] = -(W x + b - alpha * log_prior) */
int amx_ffnn_score(amx_ffnn* h, const float* feats_host, int T, float* scores_host);
int amx_ffnn_score_dev(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* scores_dev);
/* Waits for everything enqueued on the handle's stream and reports the state of the passes so far: AMX_OK, or AMX_ERR_STATE if a pass
 * of an AMX_PREC_F16MX handle met a value outside the f16 range (the scores of that pass are not valid).  The other precisions have
 * no failing pass: for them this is amx_synchronize. */
int amx_ffnn_wait_dev(amx_ffnn* h);
/* The AMX_PREC_* the handle computes in, and (nullable) the statistic that decided it.  A scorer requested as AMX_PREC_F16MX whose
 * weights are heavy-tailed computes in AMX_PREC_BF16X3 instead: with one exponent per 32 k a block whose maximum dwarfs the rest loses
 * the fp6 image of the small values (error 54-115 x that of f32 accumulation instead of 27 x; profiles/r05/f16mx_families.log).
 * *mx_block_ratio = rms of the block maxima / rms of the elements, the largest over the layers: 2.4 for Gaussian weights, 3.0 Laplace,
 * 3.3 Student-t(4), 5.0 log-normal(1.5), 5.66 at most; the switch happens above 4.0 (amx_ffnn_model.tuning mx_fallback=off keeps
 * f16mx, =auto is the default). */
int amx_ffnn_precision(const amx_ffnn* h, double* mx_block_ratio);
/* Nn::NeuralNetworkForwardNode ("neural-network-forward", Nn/Module.cc:107, Nn/NeuralNetworkForwardNode.cc:140-180): the network's
 * top-layer output per frame instead of a score.  AMX_NN_TOP_LINEAR: W x + b - alpha * log_prior (what a linear+softmax layer
 * yields with evaluate-softmax = false; exactly -score).  AMX_NN_TOP_SOFTMAX: the layer's default -- Math::FastMatrix<f32>::softmax
 * per frame (maximum, exp of the difference as ::exp(double) narrowed, sequential f32 sum, times (f32)1 / sum).  out [T x out_last]
 * in the network's own output order: a handle created with class_to_output is refused. */
enum { AMX_NN_TOP_LINEAR = 0, AMX_NN_TOP_SOFTMAX = 1 };
int amx_ffnn_forward_dev(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* out_dev, int top);
/* Same, and additionally the per-epoch statistics of amx_stats_accumulate_dev for these frames: the arg-min over
 * the states is taken in the output layer's epilogue, so the [T x n_states] score matrix is written once and
 * never re-read.  best_state_dev nullable [T]. */
int amx_ffnn_score_stats_dev(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* scores_dev,
                             uint32_t* best_state_dev, unsigned long long* state_counts_dev, double* score_sum_dev);

/* Nn::OnDemandFeatureScorer (Nn/FeatureScorer.cc:37-135): the hidden layers run once per frame
 * (forwardHiddenLayers), the output layer is evaluated only for the emissions the decoder asks for
 * (LinearAndSoftmaxLayer::getScore, Nn/LinearAndActivationLayer.cc:154-160: -bias[e] - W[e] . activation).
 * act_dev [T x amx_ffnn_hidden_dim] f32; pairs (frame_dev[p], emission_dev[p]) -> scores_dev[p]. */
int amx_ffnn_hidden_dim(const amx_ffnn* h);
int amx_ffnn_forward_hidden_dev(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* act_dev);
int amx_ffnn_score_on_demand_dev(amx_ffnn* h, const float* act_dev, int n_pairs, const uint32_t* frame_dev, const uint32_t* emission_dev,
                                 float* scores_dev);
/* Nn::PrecomputedFeatureScorer (Nn/FeatureScorer.cc:291-310): the features are network outputs computed elsewhere;
 * score(e) = -x[out(e)] + alpha * logPrior[out(e)], Core::Type<f32>::max for a disregarded class.
 * class_to_output_dev nullable (identity); log_prior_dev indexed by network output. */
int amx_precomputed_score_dev(amx_ctx* ctx, const float* feats_dev, int feats_stride, int T, int n_classes, const int* class_to_output_dev,
                              const float* log_prior_dev, float prior_scale, float* scores_dev);

/* Nn::ClassLabelWrapper::initMapping (Nn/ClassLabelWrapper.cc:56-70): classes listed in `disregard` map to -1, the others to
 * consecutive network outputs.  mapping [n_classes]; *n_targets = classes to accumulate. */
int amx_class_labels_init(int n_classes, const int* disregard, int n_disregard, int* mapping, int* n_targets);
/* Math::Vector<T> files as Math::Module's format set reads and writes them (Math/Module.cc:25-41, Core/VectorParser.hh,
 * Math/Vector.hh:286-290,357-367): XML `<vector-f32 size="n"> v ... </vector-f32>` by default, `bin:<path>` = u32 n + raw
 * elements (f32 only; the reference registers no binary format for s32).  Nn::Prior::read / write (Nn/Prior.cc:211-245) use the
 * f32 form, ClassLabelWrapper::load / save (Nn/ClassLabelWrapper.cc:72-96) the s32 form.  *data is malloc'ed (amx_free). */
int amx_nn_vector_read_f32(const char* path, int* n, float** data);
int amx_nn_vector_write_f32(const char* path, int n, const float* data);
int amx_nn_vector_read_s32(const char* path, int* n, int** data);
int amx_nn_vector_write_s32(const char* path, int n, const int* data);

/* ------------------------------------------------------------------ device buffers for resident score blocks
 * A decoder asks for ONE score at a time (Mm::FeatureScorer::ContextScorer::score, Mm/FeatureScorer.hh:31-46) and usually for a few
 * hundred of the 10^4 emissions of a frame; copying every [bufferSize x nEmissions] block to the host (40 kB per frame) would bound
 * the scorer at the PCIe rate.  These calls let a C / C++ adapter keep the block in HBM (*_score_dev) and move only what is asked
 * for: whole rows (amx_copy_to_host of one row) or (row, emission) pairs (amx_gather_scores: device gather + one small copy).
 * amx_copy_to_device returns when the source buffer may be reused (a pageable source is staged by the runtime; for a pinned source the
 * copy is waited for); amx_copy_to_host / amx_gather_scores synchronise the stream.  amx_gather_scores checks every (row, column)
 * pair against the block's shape [n_rows x ld] and refuses the call (AMX_ERR_INVALID) if one lies outside. */
int  amx_device_malloc(amx_ctx* ctx, size_t bytes, void** dev);
void amx_device_free(amx_ctx* ctx, void* dev);
int  amx_copy_to_device(amx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int  amx_copy_to_host(amx_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int  amx_gather_scores(amx_ctx* ctx, const float* scores_dev, int n_rows, int ld, int n, const uint32_t* rows_host,
                       const uint32_t* cols_host, float* dst_host);
/* Measurement: out_dev[0] = s_memtime (shader clock ticks), out_dev[1] = s_memrealtime (100 MHz), sampled by a one-wave kernel in stream
 * order.  Two samples around a stretch of work give the shader clock the chip sustained over it (delta ticks / delta 10 ns units); the
 * epoch run of bench.py prints it next to the real-time factor (Speech/CorpusProcessor.cc:49-58: wall time / audio time). */
int  amx_device_clocks_dev(amx_ctx* ctx, unsigned long long* out_dev);
/* The same per XCD: out_dev[2 x] / [2 x + 1] = (s_memtime, s_memrealtime) sampled on XCD x (eight one-wave workgroups, each filed under the
 * XCC id it reads; out_dev holds 16 values, zeroed by the caller -- an XCD no workgroup reached keeps its zeros).  The eight XCDs are
 * clocked separately.  s_memtime counts per CU and the CUs' counters are not aligned with one another: a difference of two samples is off by
 * the offset of the two CUs that took them (milliseconds' worth of ticks) -- meaningful over seconds (bench.py's full-epoch line), not over
 * one pass; s_memrealtime is one counter for the chip. */
int  amx_device_clocks_xcd_dev(amx_ctx* ctx, unsigned long long* out_dev);

/* ------------------------------------------------------------------ feature caches */

/* Core::FileArchive, the single-file "SP_ARC1" container RASR keeps feature caches, alignments and lattices in
 * (src/Core/FileArchive.cc:27-85 format, :165-225 open, :300-458 table / scan, :504-563 write; compression
 * src/Core/Archive.cc:52-222).  Host-side IO, no device involved.  AMX_ARCHIVE_WRITE opens read-write and creates the
 * file if it is missing or empty; an existing entry of the same name is replaced (the reference's
 * allow-overwrite=true).  amx_archive_close writes the file-info table (FileArchive::~FileArchive); an archive that
 * was never closed is still readable through the recovery-tag scan.  Directory and bundle archives are not handled. */
typedef struct amx_archive amx_archive;
#define AMX_ARCHIVE_READ 0
#define AMX_ARCHIVE_WRITE 1
int amx_archive_open(const char* path, int mode, amx_archive** out);
int amx_archive_close(amx_archive* a);
int amx_archive_n_files(amx_archive* a);
/* i-th live entry in archive order; *name stays valid until the archive is modified or closed. */
int amx_archive_file_info(amx_archive* a, int i, const char** name, uint32_t* size, uint32_t* compressed);
int amx_archive_has_file(amx_archive* a, const char* name);
/* Archive::readFile: *data is the uncompressed content, malloc'ed (amx_free). */
int amx_archive_read_file(amx_archive* a, const char* name, void** data, size_t* len);
/* Archive::writeFile: compress != 0 stores the gzip member the reference assembles (Archive.cc:162-215). */
int amx_archive_write_file(amx_archive* a, const char* name, const void* data, size_t len, int compress);
int amx_archive_remove_file(amx_archive* a, const char* name);

/* One Flow cache entry (= one segment) of vector-f32 packets, as Flow::CacheWriter / CacheReader exchange them
 * (src/Flow/Cache.cc:47-120, src/Flow/Datatype.cc:28-52, src/Flow/Vector.hh:88-106): blocks of
 * [string "vector-f32"][u32 n][n x (u32 dim, f32 x dim, f64 start, f64 end)].
 * feats [n x dim] row-major, times [n x 2] (start, end) -- the layout amx_gmm_score / amx_ffnn_score take.
 * gather: the node's `gather` parameter (a block holds gather+1 packets; 0xffffffff = one block per segment).
 * read: *feats / *times are malloc'ed (amx_free), times nullable; entries holding another datatype or vectors of
 * differing size return AMX_ERR_UNSUPPORTED. */
int amx_feature_cache_write(amx_archive* a, const char* segment, int n, int dim, const float* feats, const double* times,
                            unsigned gather, int compress);
int amx_feature_cache_read(amx_archive* a, const char* segment, int* n, int* dim, float** feats, double** times);
/* "<segment>.attribs": the Flow::Attributes of the cached stream as the reference's XmlWriter prints them
 * (src/Flow/Cache.cc:78-85, src/Flow/Attributes.hh:67-70,132-138).  read returns the XML text (amx_free). */
int amx_feature_cache_write_attributes(amx_archive* a, const char* segment, int n, const char* const* names,
                                       const char* const* values, int compress);
int amx_feature_cache_read_attributes(amx_archive* a, const char* segment, char** xml);

/* ------------------------------------------------------------------ NN parameter files and the state prior */

/* Binary Math::Matrix<f32> as RASR writes NN layer parameters ("bin:<base>-f32-layer-<i>.bin",
 * Nn/NeuralNetwork.cc:542-570; layout Math/Matrix.hh:560-574 + Math/Vector.hh:286-299): u32 rows, u32 cols, u32 rows,
 * then per row u32 cols + f32 values.  *data is malloc'ed row-major [rows x cols]; release it with amx_free. */
int  amx_nn_matrix_read(const char* path, int* rows, int* cols, float** data);
int  amx_nn_matrix_write(const char* path, int rows, int cols, const float* data);
void amx_free(void* p);
/* LinearLayer::setParameters (Nn/LinearLayer.cc:383-420): parameter matrix [out x (has_bias + in)], column 0 = bias,
 * -> W [out x in] row-major (== weights_[0] [in x out] column-major) and bias [out] (nullable). */
int amx_nn_layer_from_parameters(const float* params, int rows, int cols, int has_bias, float* W, float* bias);
/* Nn::Prior<f32>::setFromMixtureSet (Nn/Prior.cc:159-188), one-to-one class mapping: log_prior [n_mix]. */
int amx_prior_from_mixture_set(const amx_gmm_model* model, float* log_prior);

/* ------------------------------------------------------------------ per-epoch statistics */

/* For every frame t: best state = argmin_e scores[t][e] (first minimum wins);
 * state_counts[e] += 1 (u64), *score_sum += scores[t][best] (f64).  The caller all-reduces
 * state_counts / score_sum across ranks once per epoch. best_state_dev nullable [T]. */
int amx_stats_accumulate_dev(amx_ctx* ctx, const float* scores_dev, int T, int n_emissions,
                             uint32_t* best_state_dev, unsigned long long* state_counts_dev,
                             double* score_sum_dev);

/* ------------------------------------------------------------------ the per-epoch exchange between data-parallel ranks
 * The reference has no communication layer: `acoustic-model-trainer` processes take `partition = N`, `select-partition = k`
 * (segment i belongs to process i % N, src/Bliss/CorpusDescription.cc:174-190), write one accumulator file each, and
 * `combine-mixture-set-estimators` adds the files up offline (src/Tools/AcousticModelTrainer/AcousticModelTrainer.cc:317-325 ->
 * src/Mm/AbstractMixtureSetEstimator.cc:173-250).  Here the ranks of one node (one process and one amx_ctx per GPU) keep their
 * accumulators resident in HBM -- ONE flat f64 buffer per rank: amx_gmm_accumulator_size() doubles of statistics, score sums, and
 * the u64 counters converted with amx_counts_to_f64_dev -- and add them with ONE RCCL all-reduce over xGMI per epoch.
 *
 *   rank 0:  amx_comm_unique_id(id);  hand the 128 bytes to the other ranks (file, environment, MPI, socket: the caller's business)
 *   all:     amx_comm_init(ctx, rank, world, id, &comm);          collective: returns when all `world` ranks have called it
 *            ... one epoch of amx_*_score_stats_dev / amx_gmm_accumulate_dev into the flat buffer ...
 *            amx_comm_all_reduce_f64_dev(comm, flat_dev, n);      in place, on the context's stream, sum over ranks
 *
 * RCCL is bound at run time (dlopen of librccl.so; AMX_RCCL_LIB names another file): the library loads and every other entry
 * point works without it, amx_comm_available() tells.  The result equals the single-process accumulators up to the f64 summation
 * order of the ring.  A communicator belongs to its context (same device, same stream) and must be destroyed before it. */
typedef struct amx_comm amx_comm;
#define AMX_COMM_ID_BYTES 128
int  amx_comm_available(void);
int  amx_comm_unique_id(unsigned char id[AMX_COMM_ID_BYTES]);
int  amx_comm_init(amx_ctx* ctx, int rank, int world, const unsigned char id[AMX_COMM_ID_BYTES], amx_comm** out);
int  amx_comm_rank(const amx_comm* c);
int  amx_comm_world(const amx_comm* c);
int  amx_comm_all_reduce_f64_dev(amx_comm* c, double* buf_dev, size_t n);
void amx_comm_destroy(amx_comm* c);
/* Integer counters (state counts: u64, Mm/AbstractMixtureSetEstimator.cc keeps them as Weight = f64 anyway) ride in the same f64
 * buffer: exact below 2^53.  counts -> doubles before the all-reduce, doubles -> counts (rounded to nearest) after it. */
int amx_counts_to_f64_dev(amx_ctx* ctx, const unsigned long long* counts_dev, double* out_dev, size_t n);
int amx_f64_to_counts_dev(amx_ctx* ctx, const double* in_dev, unsigned long long* counts_dev, size_t n);

#ifdef __cplusplus
}
#endif
#endif
