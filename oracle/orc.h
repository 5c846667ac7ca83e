/*
 * oracle/orc.h -- CPU restatement of the RASR acoustic front-end + emission scorers.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so.  The product (rasr_amd/) never
 * links, imports or calls anything in this directory.
 *
 * Plain C, no dependency on the reference tree.  Every function cites the reference
 * file:line whose arithmetic it restates (paths relative to /root/reference/src).
 *
 * THE REFERENCE HAS TWO ARITHMETICS, and so has this library (cmake_resources/CompileOptions.cmake:21-48):
 *   contract=off  liboracle.so      the reference configured with -DMARCH=x86-64 (or built on a host without FMA units): -msse3 only,
 *                                   every f32 / f64 operation rounds once.
 *   contract=fma  liboracle_fma.so  the reference's DEFAULT configuration (MARCH defaults to "native") on any FMA host: GCC's default
 *                                   -ffp-contract=fast (the build uses gnu++20) fuses a product whose only use is an addition or
 *                                   subtraction into one fused multiply-add.  The same sources with -DORC_CONTRACT_FMA: the sites
 *                                   where GCC contracts are written ORC_FMAF / ORC_FMA (everything else is compiled with
 *                                   -ffp-contract=off in BOTH flavours, so the flavours differ at exactly those sites).
 * Which sites: read off the reference built both ways (oracle/ref/Makefile: libref.so / libref_native.so, objdump of the native
 * objects) -- the GMM distance's `sum += df * df` (vfmadd231ps / vfmadd231ss, function-text pin gdm_distance), the f32 dot product
 * of Math::Vector (cosine transform), gaussLogNormFactor's N * log(2 pi) + sum (f64), the filter bank's apply, the regression
 * sums and the batch-float scorer's SSE accumulate (function-text pins filter_apply, regression, batch_float_fill).  Sites in
 * translation units that cannot be compiled here follow GCC's rule by reading (the other back-end sums) and say so; preemphasis'
 * `v[i] -= alpha * previous` is pinned too (function-text pin preemphasis).
 * NOT restated in fma form: the FFT's f64 twiddle recurrences (14 fused operations in the native object; the f32 results were
 * bit-identical to the plain build on every frame tried, tests/test_contract.py) and the f4 front ends / quantised scorers.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   FFT core, framing/flush, mel warp/derivative/inverse, GMM logNorm / 1/sqrt(var):
 *       pinned bit-exactly against oracle/_ref (reference sources compiled unmodified).
 *   preemphasis, Hamming table, the bank's boundary (filter count, width, spacing, centres) and one filter (interval, weights, apply),
 *   GMM distance, regression, batch-float sum / minimum:
 *       pinned bit-exactly on the reference's
 *       own function text compiled with both flag sets (oracle/ref/extract_fn.py, tests/test_contract.py).
 *   GMM max score (combine / tie rule) and the log-add scorer: the reference's calculateScoreAndDensity text, both builds (extract_fn.py
 *       gdm_distance).  The filter bank as a whole: pinned by the known answers the
 *       reference produced in this container.
 *   NN forward: pinned by the reference's own unit-test vectors
 *       (Test/Nn_LinearAndActivationLayer.cc, Test/Nn_NeuralNetwork.cc).
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* a * b + c as the reference's build evaluates it at a site GCC contracts (see the header comment) */
#ifdef ORC_CONTRACT_FMA
#define ORC_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
#define ORC_FMA(a, b, c) __builtin_fma((a), (b), (c))
#else
#define ORC_FMAF(a, b, c) ((a) * (b) + (c))
#define ORC_FMA(a, b, c) ((a) * (b) + (c))
#endif
/* 0 = this library restates the contract=off build, 1 = contract=fma */
int orc_contract(void);

/* ---------------------------------------------------------------- MFCC chain */

typedef struct {
    double sample_rate;            /* Hz, e.g. 16000 */
    double win_len_s;              /* signal-window length, mfcc.flow: 0.025 */
    double win_shift_s;            /* signal-window shift,  mfcc.flow: 0.01  */
    double preemph_alpha;          /* signal-preemphasis alpha, mfcc.flow: 1.00 */
    double fft_max_input_s;        /* maximum-input-size, mfcc.flow: 0.025 */
    int    apply_scale;            /* apply-scale (default true): multiply by 1/fs */
    double mel_filter_width;       /* filter-width, default 268.258 */
    double mel_spacing;            /* spacing, default 0 -> 0.5*width */
    int    warp_differential_unit; /* default true */
    int    n_ceps;                 /* nr-outputs of signal-cosine-transform */
    int    dct_normalize;          /* normalize (default false) */
    /* front_end 1 = mfplp.flow (Tools/FeatureExtraction/share/mfplp.flow): power spectrum -> mel filter bank -> ^plp_power ->
     * cosine transform (N-plus-one input, nr-outputs n_autocorrelation, normalize) -> Levinson -> LPC cepstrum (n_ceps) */
    int    front_end;              /* 0 = mfcc.flow, 1 = mfplp.flow */
    int    n_autocorrelation;      /* nr-autocorrelation-coefficients (LPC order + 1) */
    double plp_power;              /* intensity-loudness-law value, mfplp.flow: 0.33 */
    /* signal-filterbank parameters beyond mfcc.flow's (Signal/Filterbank.cc:700-745) and plp.flow
     * (Tools/FeatureExtraction/share/plp.flow): front_end 2 = Hamming 20 ms, no preemphasis node (alpha 0), power spectrum ->
     * trapeze / include-boundary / bark filter bank (width 3.8, spacing 0.93853) -> first and last output duplicated ->
     * equal-loudness preemphasis (multiplies) -> ^plp_power -> cosine transform (N-plus-one) -> Levinson -> LPC cepstrum */
    int    filter_type;            /* type: 0 triangular, 1 trapeze */
    int    boundary;               /* boundary: 0 stretch-to-cover, 1 include-boundary, 2 emphasize-boundary */
    int    warping;                /* warping-function: 0 mel, 1 bark */
} orc_mfcc_cfg;

typedef struct orc_mfcc orc_mfcc;

orc_mfcc* orc_mfcc_create(const orc_mfcc_cfg* cfg);
void      orc_mfcc_destroy(orc_mfcc* h);

/* geometry */
int  orc_mfcc_frame_len(const orc_mfcc* h);   /* samples per window (400) */
int  orc_mfcc_frame_shift(const orc_mfcc* h); /* 160 */
int  orc_mfcc_fft_len(const orc_mfcc* h);     /* 512 */
int  orc_mfcc_n_bins(const orc_mfcc* h);      /* 257 */
int  orc_mfcc_n_filters(const orc_mfcc* h);   /* 20 / 40 */
int  orc_mfcc_n_ceps(const orc_mfcc* h);
long orc_mfcc_n_frames(const orc_mfcc* h, long n_samples);

/* tables (pointers owned by h) */
const float* orc_mfcc_window(const orc_mfcc* h);          /* [frame_len] */
const int*   orc_mfcc_filter_start(const orc_mfcc* h);    /* [n_filters] */
const int*   orc_mfcc_filter_end(const orc_mfcc* h);      /* [n_filters] */
const int*   orc_mfcc_filter_offset(const orc_mfcc* h);   /* [n_filters+1] into weights */
const float* orc_mfcc_filter_weights(const orc_mfcc* h);  /* concatenated */
const float* orc_mfcc_dct(const orc_mfcc* h);             /* [n_ceps][n_filters] row-major; plp.flow: [n_autocorrelation][n_filters + 2] */
const double* orc_mfcc_equal_loudness(const orc_mfcc* h); /* plp.flow: [n_filters + 2], else NULL */
int    orc_core_is_almost_equal(double a, double b, double tolerance);          /* Core/Utility.hh:322-327 */
int    orc_core_is_significantly_greater(double a, double b, double tolerance); /* Core/Utility.hh:343-345 */
double orc_bark(double f);
double orc_bark_derivative(double f);
double orc_bark_inverse(double b);
double orc_equal_loudness(double f);
double orc_equal_loudness_4khz(double f);
double       orc_mfcc_mel_max(const orc_mfcc* h);         /* warped maximum frequency */

/* ---------------------------------------------------------------- gammatone front-end (orc_gammatone.c; filter bank, windows, temporal and
 * spectral integration pinned on the reference's function text in both builds) */
typedef struct {
    double sample_rate;
    int    cascade;          /* signal-gammatone cascade (node default 4) */
    double min_freq;         /* minfreq (100) */
    double max_freq;         /* maxfreq (6000) */
    double q;                /* q (9.264491981582191) */
    int    channels;         /* channels (50) */
    int    cf_mode;          /* cfmode: 0 human, 1 erb */
    double warp_freq_break;  /* warp-freqbreak (6600) */
    double warping_factor;   /* warping-factor (1) */
    int    ti_window;        /* signal-temporalintegration type: 0 hanning, 1 rectangular */
    double ti_length_s;      /* length */
    double ti_shift_s;       /* shift */
    int    si_window;        /* signal-spectralintegration type: 0 hanning, 1 rectangular */
    int    si_length;        /* length in channels; 0 = node absent */
    int    si_shift;         /* shift in channels */
    double power;            /* generic-vector-f32-power value; 0 = node absent */
    int    n_ceps;           /* signal-cosine-transform nr-outputs; 0 = node absent */
    int    dct_normalize;
} orc_gammatone_cfg;
float orc_window_value(int type, int len, int i); /* value i of the integration nodes' window: type 0 Hanning, 1 rectangular */
void  orc_temporal_integrate(int window, const float* frame, int rows, int channels, float* out); /* TemporalIntegration::transform, one frame */
int   orc_spectral_integrate(const float* win, int length, int shift, const float* in, int channels, float* out); /* SpectralIntegration::apply, one row */
typedef struct orc_gammatone orc_gammatone;
orc_gammatone* orc_gammatone_create(const orc_gammatone_cfg* cfg);
void           orc_gammatone_destroy(orc_gammatone* h);
int            orc_gammatone_n_out(const orc_gammatone* h);
int            orc_gammatone_frame_len(const orc_gammatone* h);
int            orc_gammatone_frame_shift(const orc_gammatone* h);
int            orc_gammatone_si_channels(const orc_gammatone* h);
const float*   orc_gammatone_center_frequencies(const orc_gammatone* h);
const float*   orc_gammatone_coefficients(const orc_gammatone* h); /* [channels][4] a0 a1 b1 b2 */
long           orc_gammatone_n_frames(const orc_gammatone* h, long n_samples);
long           orc_time_window_frames(long n, int length, int shift, long* starts, int* lens, long cap);
long long      orc_dc_detection(const float* pcm, long long n, long long block, double sample_rate, double min_dc_length_s, float max_dc_increment,
                                double min_non_dc_segment_length_s, int maximal_output_size, long long* starts, long long* lens, long long cap);
/* filtered [n_samples x channels] (nullable): the signal-gammatone output; out [n_frames x n_out] */
long           orc_gammatone_run(const orc_gammatone* h, const float* pcm, long n_samples, float* filtered, float* out);

/* whole utterance: pcm f32 (s16 values, unscaled) -> ceps [n_frames x n_ceps] row-major.
 * returns number of frames written. */
long orc_mfcc_run(const orc_mfcc* h, const float* pcm, long n_samples, float* ceps);

/* per-stage taps for one frame of an utterance (any pointer may be NULL) */
int orc_mfcc_stages(const orc_mfcc* h, const float* pcm, long n_samples, long frame,
                    float* windowed /*[fft_len], zero padded*/,
                    float* spectrum /*[fft_len+2] alternating re,im, scaled*/,
                    float* amplitude /*[n_bins]*/, float* mel /*[n_filters]*/,
                    float* logmel /*[n_filters]*/, float* ceps /*[n_ceps]*/);

/* building blocks, exposed so they can be pinned one by one */
/* Math::LevinsonLeastSquares::work + gain() + a() (Math/LevinsonLse.cc:35-70, LevinsonLse.hh:49-66): R [n] autocorrelation,
 * a [n-1]; returns 0 when the recursion meets a zero prediction error (the reference reports an error for the frame) */
int    orc_levinson(const float* R, int n, float* gain, float* a);
/* Signal::autoregressionToCepstrum (Signal/AutoregressionToCepstrum.cc:21-35): c [nc], 2 <= nc <= na + 1 */
void   orc_ar_to_cepstrum(float gain, const float* a, int na, float* c, int nc);
void   orc_preemphasis(float* x, long n, float alpha);            /* in place, segment start */
void   orc_hamming_window(float* w, int len); /* Signal/WindowFunction.cc:92-101: the table (len <= 1: zeros, the reference's init() fails) */
void   orc_cosine_table(int n_plus_one, int rows, int cols, float* table);  /* CosineTransform::initEvenAboutNminusHalf / initNplusOneData */
void   orc_cosine_transform(int n_plus_one, int n_in, int n_out, int normalize, const float* in, float* out, float* table_out);
int    orc_filter_boundary(int type, double width, double spacing, double ncp, double fmin, double fmaxw, double* width_out,
                           double* spacing_out, double* centers, int cap); /* number of filters, final width / spacing, centres */
double orc_filter_center(int type, size_t i, double width, double spacing, double ncp, double fmin);
int    orc_filter_build(int type, int warping, double center, double width, double fmin, double fmaxw, double d2c, int diff, int* start,
                        int* end, float* weights, int cap); /* FilterBuilder::create for one filter: its interval and weights */
float  orc_filter_apply(const float* in, int start, int end, const float* weights); /* one filter of the bank: sum over bins [start, end) */
void   orc_fft_real(float* v, int n);                             /* Math::FastFourierTransform::transformReal */
void   orc_fft_complex(float* v, int n_floats);                   /* ::transform (forward) */
double orc_mel(double f);                                         /* continuous-domain mel warp */
double orc_mel_derivative(double f);
double orc_mel_inverse(double m);

/* ---------------------------------------------------------------- diag-GMM */

typedef struct {
    int dim, n_mix, n_dens, n_mean, n_cov;
    const uint32_t* mix_offsets; /* [n_mix+1] */
    const uint32_t* dens_index;  /* [sum K_m] density index per (mixture, k) */
    const double*   log_weight;  /* [sum K_m] f64 log weights (Mm::Weight) */
    const uint32_t* dens_mean;   /* [n_dens] */
    const uint32_t* dens_cov;    /* [n_dens] */
    const float*    means;       /* [n_mean x dim] */
    const float*    variances;   /* [n_cov  x dim] */
    double          mixture_weight_scale;
    double          gaussian_scale;
} orc_gmm_model;

typedef struct orc_gmm orc_gmm;
orc_gmm* orc_gmm_create(const orc_gmm_model* m);
void     orc_gmm_destroy(orc_gmm* h);
/* prepared tables, for pinning */
const float* orc_gmm_minus2_log_weights(const orc_gmm* h); /* [sum K_m] */
const float* orc_gmm_inv_sqrt_var(const orc_gmm* h);       /* [n_cov x dim] */
const float* orc_gmm_log_norm(const orc_gmm* h);           /* [n_cov] */
/* mode 0 = maximum approximation, 1 = log-add.  feats [T x dim] row-major;
 * scores [T x n_mix]; best [T x n_mix] density-in-mixture index (nullable) */
void orc_gmm_score(const orc_gmm* h, int mode, const float* feats, int T, float* scores, uint32_t* best);
/* GaussDiagonalMaximumFeatureScorer::distance alone (pinned by tests/test_contract.py on the reference's own function text) */
float orc_gmm_distance(const float* x, const float* mu, const float* inv_sqrt_var, int dim);
/* Mm::BatchFloatFeatureScorer arithmetic (pooled covariance only, n_cov == 1) */
int orc_gmm_score_preselection_float(const orc_gmm* h, const double* log_weight, const float* variances, const float* feats, int T,
                                     int n_clusters, int n_select, int iterations, float backoff, float* scores,
                                     uint32_t* cluster_of_out, float* cluster_means_out, int* n_clusters_out);
/* Mm::BatchPreselectionIntFeatureScorer ("preselection-batch-int"): cluster_means [n_clusters x dim] u8; its clustering is pinned on the
 * reference's template text (orc_cluster_u8), the integer scoring loop is read */
int orc_gmm_score_preselection_int(const orc_gmm* h, const double* log_weight, const float* variances, const float* feats, int T,
                                   int n_clusters, int n_select, int iterations, float* scores, uint32_t* cluster_of_out,
                                   uint8_t* cluster_means_out, int* n_clusters_out);
void  orc_cluster_u8(const uint8_t* means, int nk, int dim, int n_clusters, int iterations, uint32_t* cof, uint8_t* cm); /* DensityClustering<u8, s32> */
void  orc_cluster_select(const float* cm, int n_clusters, int pdim, int n_select, const float* xs, unsigned char* sel); /* DensityClustering::selectClusters */
float orc_batch_float_fill(const float* ms, const float* cst, int nk, const float* xs, int pdim); /* fillScoreCacheTpl, one feature x one mixture */
int orc_gmm_score_batch_float(const orc_gmm* h, const double* log_weight, const float* variances,
                              const float* feats, int T, float* scores);

/* Mm::SimdGaussDiagonalMaximumFeatureScorer ("SIMD-diagonal-maximum"): u8-quantised means and features, integer distance;
 * scaling_out (nullable) receives the quantisation scaling factor */
int      orc_gmm_score_simd(const orc_gmm* h, const double* log_weight, const float* variances, const float* feats, int T,
                            float* scores, uint32_t* best, float* scaling_out);
unsigned orc_quantize(float v);
/* Mm::BatchIntFeatureScorer ("batch-diagonal-maximum-int" / "-fast"): pooled covariance only (-1 otherwise) */
int      orc_gmm_score_batch_int(const orc_gmm* h, const double* log_weight, const float* variances, const float* feats, int T,
                                 float* scores);

/* Viterbi training statistics: see orc_score.c for the accumulator layout */
long orc_gmm_accumulator_size(const orc_gmm* h);
void orc_gmm_accumulate(const orc_gmm* h, const float* feats, int T, const uint32_t* mixture,
                        const uint32_t* density_in_mixture, double* acc);
/* weighted Viterbi (mode 0) / Baum-Welch (mode 1) statistics, Mm/AbstractMixtureSetEstimator.cc:127-147; weight nullable (= 1) */
void orc_gmm_accumulate_weighted(const orc_gmm* h, int mode, const float* feats, int T, const uint32_t* mixture,
                                 const double* weight, const uint32_t* density_in_mixture, double* acc);

/* ---------------------------------------------------------------- FFNN forward */

enum { ORC_ACT_NONE = 0, ORC_ACT_RELU = 1, ORC_ACT_SIGMOID = 2, ORC_ACT_TANH = 3 };

typedef struct {
    int                 n_layers;
    const int*          in_dim;
    const int*          out_dim;
    const float* const* W;    /* per layer [out x in] row-major == RASR weights_[0] [in x out] col-major */
    const float* const* bias; /* per layer [out] */
    const int*          activation;
    const float*        log_prior; /* nullable [out_last] */
    float               prior_scale;
} orc_ffnn_model;

/* feats [T x in0] row-major; scores [T x out_last] = -(W x + b - alpha*logprior).
 * acc64: 0 = f32 multiply-then-add in ascending k, 1 = f64 accumulation (tight "truth"),
 * 2 = f32 fmaf chain in ascending k (bit pattern of an f32 MFMA / FMA GEMM). */
float orc_activation(float v, int act); /* one activation value (ORC_ACT_*), as the layers apply it */
void orc_ffnn_score(const orc_ffnn_model* m, const float* feats, int T, float* scores, int acc64);
void orc_softmax_rows(float* x, int T, int n);
void orc_ffnn_forward(const orc_ffnn_model* m, const float* feats, int T, float* out, int top, int acc64);

#ifdef __cplusplus
}
#endif

/* ---- feature back-end, orc_backend.c; normalisation (five types), regression and matrix multiplication pinned on
 * the reference's text / headers in both builds (see that file) */
void orc_normalize(const float* in, int n, int dim, int type, int length, int right, float* out);
void orc_normalize_ex(const float* in, int n, int dim, int type, int level, int length, int right, float* out);
void orc_regression(const float* in, int n, int dim, int order, int right, float* out);
void orc_matrix_multiply(const float* M, int rows, int cols, const float* in, int T, float* out);
void orc_vector_normalize(int type, const float* in, int n, int dim, float* out);
void orc_vector_function(int kind, float prm, const float* in, long n, int dim, float* out);

#endif
