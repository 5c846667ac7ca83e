/*
 * oracle/orc_mfcc.c -- CPU restatement of the mfcc.flow chain (TEST INFRASTRUCTURE, see orc.h).
 *
 *   signal-preemphasis -> signal-window(hamming) -> signal-real-fast-fourier-transform
 *   -> signal-vector-alternating-complex-f32-amplitude -> signal-filterbank(mel)
 *   -> generic-vector-f32-log -> signal-cosine-transform
 *   (Tools/FeatureExtraction/share/mfcc.flow:8-34)
 *
 * and, with cfg.front_end = 1, of mfplp.flow (same share directory, lines 9-47):
 *   ... amplitude -> generic-vector-f32-power(2) -> signal-filterbank(mel) -> generic-vector-f32-power(0.33)
 *   -> signal-cosine-transform(N-plus-one, normalize) -> signal-autocorrelation-to-autoregression
 *   -> signal-autoregression-to-cepstrum
 *
 * Arithmetic types follow the reference exactly: f32 sample data, f64 trigonometric
 * recurrences and table construction, f32 tables.  Compile with -ffp-contract=off.
 */
#include "orc.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct orc_mfcc {
    orc_mfcc_cfg cfg;
    int          frame_len, frame_shift, fft_len, n_bins, n_filters, n_ceps;
    float        fft_scale;    /* 1/(f32)fs */
    float*       window;       /* [frame_len] */
    int *        f_start, *f_end, *f_off;
    float*       f_weights;
    float*       dct;          /* [n_ceps][n_filters]; MF-PLP: [n_autocorrelation][n_filters], N-plus-one input type;
                                * PLP: [n_autocorrelation][n_filters + 2] */
    double       mel_max;
    double*      eql;          /* PLP: equal-loudness factor per element of the first/last-extended filter-bank vector [n_filters + 2] */
};

/* ------------------------------------------------------------------ preemphasis
 * Signal/Preemphasis.cc:51-77.  At segment start previous_ = x[0].  alpha == 1 uses the
 * pure first difference; otherwise y[i] = x[i] - alpha*prev with an f32 product. */
void orc_preemphasis(float* x, long n, float alpha) {
    if (n <= 0)
        return;
    float prev = x[0];
    if (alpha != 1.0) {
        for (long i = 0; i < n; ++i) {
            float cur  = x[i];
            x[i]       = ORC_FMAF(-alpha, prev, cur); /* v[i] -= alpha_ * previous_: vfnmadd132ss in the default build (function-text pin preemphasis) */
            prev       = cur;
        }
    }
    else {
        for (long i = n - 1; i > 0; --i)
            x[i] = x[i] - x[i - 1];
        x[0] = x[0] - prev;
    }
}

/* ------------------------------------------------------------------ FFT
 * Math/FastFourierTransform.cc:22-23 -- note DPi is a truncated 2*pi. */
static const double ORC_PI  = 3.141592653589793238;
static const double ORC_DPI = 6.28318530717959;

/* Math/FastFourierTransform.cc:28-57: bit reversal of the n_floats/2 complex values */
static void orc_bit_reverse(float* v, int n_floats) {
    int half = n_floats / 2;
    int j    = 1;
    for (int i = 1; i < n_floats; i += 2) {
        if (j > i) {
            float a = v[i - 1], b = v[i];
            v[i - 1] = v[j - 1];
            v[i]     = v[j];
            v[j - 1] = a;
            v[j]     = b;
        }
        int m = half;
        while (m >= 2 && j > m) {
            j -= m;
            m >>= 1;
        }
        j += m;
    }
}

/* Math/FastFourierTransform.cc:59-93: radix-2 DIT, +i sign, f64 twiddle recurrence,
 * twiddle*data product formed in f64 and rounded to f32, butterflies in f32. */
void orc_fft_complex(float* v, int n_floats) {
    orc_bit_reverse(v, n_floats);
    for (int span = 2; span < n_floats; span <<= 1) {
        int    stride = span << 1;
        double theta  = ORC_DPI / span;
        double sh     = sin(0.5 * theta);
        double dr     = -2.0 * sh * sh;
        double di     = sin(theta);
        double wr = 1.0, wi = 0.0;
        for (int m = 1; m < span; m += 2) {
            for (int i = m; i <= n_floats; i += stride) {
                int   j  = i + span;
                float tr = (float)(wr * v[j - 1] - wi * v[j]);
                float ti = (float)(wr * v[j] + wi * v[j - 1]);
                v[j - 1] = v[i - 1] - tr;
                v[j]     = v[i] - ti;
                v[i - 1] += tr;
                v[i] += ti;
            }
            double old = wr;
            wr         = wr * dr - wi * di + wr;
            wi         = wi * dr + old * di + wi;
        }
    }
}

/* Math/FastFourierTransform.cc:95-146 (forward branch): complex transform of the packed
 * data followed by the even/odd split with f64 temporaries. */
void orc_fft_real(float* v, int n) {
    const double theta = ORC_PI / (n >> 1);
    const float  c     = -0.5f;
    orc_fft_complex(v, n);
    double sh = sin(0.5 * theta);
    double dr = -2.0 * sh * sh;
    double di = sin(theta);
    double wr = dr + 1;
    double wi = di;
    for (int i = 1; i < (n >> 2); ++i) {
        int    a = i + i, b = a + 1, p = n - a, q = p + 1;
        double h1r = 0.5 * (v[a] + v[p]);
        double h1i = 0.5 * (v[b] - v[q]);
        double h2r = -c * (v[b] + v[q]);
        double h2i = c * (v[a] - v[p]);
        v[a]       = (float)(h1r + wr * h2r - wi * h2i);
        v[b]       = (float)(h1i + wr * h2i + wi * h2r);
        v[p]       = (float)(h1r - wr * h2r + wi * h2i);
        v[q]       = (float)(-h1i + wr * h2i + wi * h2r);
        double old = wr;
        wr         = wr * dr - wi * di + wr;
        wi         = wi * dr + old * di + wi;
    }
    float h = v[0];
    v[0]    = h + v[1];
    v[1]    = h - v[1];
}

/* ------------------------------------------------------------------ mel warping
 * Math/AcousticalAnalyticFunctions.hh:24-60 composed as in
 * Math/AnalyticFunctionFactory.cc:338-341 (continuous domain): nest(scale(2595), melCore). */
double orc_mel(double f) {
    return 2595.0 * log10(1.0 + f / 700.0);
}
/* derive(): AnalyticNesting::derive (Math/AnalyticFunction.hh:119-122) gives
 * (const(2595) o melCore)(f) * derivedMelCore(f) */
double orc_mel_derivative(double f) {
    return 2595.0 * (1.0 / log(10) / (700.0 + f));
}
/* invert(): nest(inverseMelCore, scale(1/2595)) (Math/AnalyticFunction.hh:123-126,
 * Math/SimpleAnalyticFunctions.hh:107-123) */
double orc_mel_inverse(double m) {
    double a = 1 / 2595.0;
    return (pow(10, a * m) - 1.0) * 700.0;
}

/* ------------------------------------------------------------------ bark warping
 * Math/AnalyticFunctionFactory.cc:369-373 (continuous domain): nest(scaling(6), nest(asinh, scaling(1 / 600)))
 * with Math/SimpleAnalyticFunctions.hh:107-123 (ScalingFunction), :158-176,214-222 (ArcSinh, DerivedArcSinh, Sinh). */
double orc_bark(double f) {
    return 6.0 * asinh((1.0 / 600.0) * f);
}
/* AnalyticNesting::derive twice: (const(6) o g)(f) * g'(f), g' = (DerivedArcSinh o scaling)(f) * const(1 / 600) */
double orc_bark_derivative(double f) {
    double u = (1.0 / 600.0) * f;
    return 6.0 * (((double)1 / sqrt(ORC_FMA(u, u, (double)1))) * (1.0 / 600.0)); /* DerivedArcSinh::value: vfmadd132sd in the default build */
}
/* AnalyticNesting::invert twice: scaling(1 / (1 / 600)) o sinh o scaling(1 / 6) */
double orc_bark_inverse(double b) {
    double s6 = 1 / 6.0, s600 = 1 / (1.0 / 600.0);
    return s600 * sinh(s6 * b);
}

/* Math/AcousticalAnalyticFunctions.cc:21-37 */
double orc_equal_loudness(double f) {
    double omega       = 2 * M_PI * f;
    double omegaSquare = omega * omega;
    double omegaFourth = omegaSquare * omegaSquare;
    double omegaSixth  = omegaFourth * omegaSquare;
    return (omegaFourth * (omegaSquare + 56.8e6)) /
           ((omegaSquare + 6.3e6) * (omegaSquare + 6.3e6) * (omegaSquare + 0.38e9) * (omegaSixth / 9.58e26 + 1));
}
double orc_equal_loudness_4khz(double f) {
    double omega       = 2 * M_PI * f;
    double omegaSquare = omega * omega;
    double termFourth  = omegaSquare / (omegaSquare + (double)6.3e6);
    return termFourth * termFourth * (omegaSquare + (double)56.8e6) / (omegaSquare + (double)0.38e9);
}

/* Flow attributes carry doubles as text with 6 significant digits
 * (Flow/Attributes.hh:109-113: ostringstream << f64), and nodes read them back with atof. */
static double orc_attr_roundtrip(double x) {
    char buf[64];
    snprintf(buf, sizeof buf, "%g", x);
    return atof(buf);
}

static int orc_almost_integer(double x) { /* Signal/Filterbank.cc:691-694 */
    return fabs(x - round(x)) < 1e-10;
}

/* Core/Utility.hh:322-327 */
static int orc_almost_equal_tol(double a, double b, double tolerance) {
    const double eps   = 2.220446049250313e-16; /* Core::Type<f64>::epsilon = DBL_EPSILON */
    const double delta = 2.2250738585072014e-308; /* Core::Type<f64>::delta   = DBL_MIN */
    double       d     = fabs(a - b);
    double       e     = (fabs(a) + fabs(b) + delta) * eps * tolerance;
    return d < e;
}
static int orc_almost_equal(double a, double b) {
    return orc_almost_equal_tol(a, b, 1.0);
}
/* for pinning against Core::isAlmostEqual / Core::isSignificantlyGreater (tests only) */
int orc_core_is_almost_equal(double a, double b, double tolerance) { return orc_almost_equal_tol(a, b, tolerance); }
int orc_core_is_significantly_greater(double a, double b, double tolerance) { return a > b && !orc_almost_equal_tol(a, b, tolerance); }

/* ------------------------------------------------------------------ table construction */

/* Signal/WindowFunction.cc:92-101 (Hamming), symmetric fill, f64 -> f32.  PINNED on the reference's function text in both builds
 * (oracle/ref/extract_fn.py hamming_window): the default build fuses 0.54 - 0.46 * cos() into one vfnmadd132sd, and the f32 table has
 * the same bits for every length 2 .. 4096 -- no contract form needed here (tests/test_contract.py). */
static void orc_build_hamming(float* w, int len) {
    if (len <= 1) {
        for (int i = 0; i < len; ++i)
            w[i] = 0; /* init() fails in the reference; not reachable from mfcc.flow */
        return;
    }
    unsigned M = (unsigned)len - 1;
    for (unsigned n = 0; n <= M / 2; ++n) {
        float c  = (float)(0.54 - 0.46 * cos(2.0 * M_PI * n / M));
        w[n]     = c;
        w[M - n] = c;
    }
}

void orc_hamming_window(float* w, int len) { orc_build_hamming(w, len); }

/* Signal/Filterbank.cc:144-244 (filter builder, triangle), :246-275 (trapeze), :330-470 (boundaries: include-boundary),
 * :519-567 (stretch-to-cover), :575-595 (emphasize-boundary), :640-672 (FilterBank::init: with warp-center-positions = true, the
 * default, the boundary works on the warped axis with an identity warping), :765-819 (node init); warping of the continuous
 * frequency axis by mel or bark. */
typedef struct {
    double (*value)(double);
    double (*derivative)(double);
    double (*inverse)(double);
} orc_warp;

/* TrapezeFilterBuilder (Signal/Filterbank.cc:246-275) */
static float orc_trapeze_weight(double frequency, double center, double width) {
    const double nmb               = 0.5 / (1.3 - (-2.5)); /* normalizedMiddleBorder */
    double       relativeFrequency = frequency - center;
    double       middleLeftBorder  = -nmb * width;
    if (relativeFrequency < middleLeftBorder)
        return (float)pow(10, relativeFrequency - middleLeftBorder);
    double middleRightBorder = nmb * width;
    if (relativeFrequency <= middleRightBorder)
        return 1;
    return (float)pow(10, -2.5 * (relativeFrequency - middleRightBorder));
}

static double orc_postprocess_nfilters(double nf) { /* Boundary::postprocessNumberOfFilters */
    if (nf < 1)
        return 1;
    if (orc_almost_integer(nf))
        return round(nf);
    return nf;
}

/* ONE filter (Signal/Filterbank.cc:144-217 FilterBuilder::create / setStart / setEnd / setWeights, :236-244 and :268-281 the
 * triangular and the trapeze weight()): type 0 triangular / 1 trapeze, warping 0 mel / 1 bark, d2c = scaling of the discrete axis.
 * Returns the number of weights (end - start), -1 where the reference's builder fails, -2 if `cap` is too small.  PINNED on the
 * reference's function text in both builds (oracle/ref/extract_fn.py filter_build, tests/test_contract.py). */
int orc_filter_build(int type, int warping, double center, double width, double fmin, double fmaxw, double d2c, int diff, int* start_out,
                     int* end_out, float* weights, int cap) {
    const orc_warp mel = {orc_mel, orc_mel_derivative, orc_mel_inverse}, bark = {orc_bark, orc_bark_derivative, orc_bark_inverse};
    const orc_warp* W       = warping == 1 ? &bark : &mel;
    double          inv_d2c = 1 / d2c; /* ScalingFunction::invert */
    double          ncp     = type == 1 ? 2.5 / (1.3 - (-2.5)) : 0.5;
    /* setStart */
    double lo = ORC_FMA(-ncp, width, center); /* center_ - normalizedCenterPosition() * width_: vfnmadd231sd in the default build */
    if (!(lo > fmin))
        lo = fmin; /* std::max(a, b) returns a unless a < b */
    double s = inv_d2c * W->inverse(lo);
    s        = orc_almost_integer(s) ? round(s) : ceil(s);
    if (!(s >= 0))
        return -1;
    /* setEnd */
    double hi = ORC_FMA(1.0 - ncp, width, center); /* vfmadd132sd */
    if (fmaxw < hi)
        hi = fmaxw;
    double e = inv_d2c * W->inverse(hi);
    e        = orc_almost_integer(e) ? round(e) + 1 : ceil(e);
    size_t start = (size_t)s;
    if (!(e > 0 && start < (size_t)e))
        return -1;
    size_t end = (size_t)e;
    if (end - start > (size_t)cap)
        return -2;
    *start_out = (int)start;
    *end_out   = (int)end;
    /* setWeights: f32 shape weight times f64 derivative, rounded to f32 */
    int n = 0;
    for (unsigned b = (unsigned)start; b < end; ++b) {
        double fw = W->value(d2c * (double)b);
        float  sh;
        if (type == 1)
            sh = orc_trapeze_weight(fw, center, width);
        else {
            sh = (float)((double)1 - fabs(fw - center) / (width / 2));
            if (!(sh >= 0))
                sh = 0;
        }
        double der   = diff ? W->derivative(d2c * (double)b) : 1.0;
        weights[n++] = (float)(sh * der);
    }
    return n;
}

/* The boundary of a bank (Signal/Filterbank.cc:428-470 Boundary::init / setSpacing / postprocessNumberOfFilters, :495-501
 * IncludeBoundary::getNumberOfFilters, :534-540 and :546-567 StretchToCover, :584-589 EmphasizeBoundary; FilterBank::init :640-650 with
 * warp-center-positions = true: the boundary works on the warped axis with an identity warping): type 0 stretch-to-cover / 1
 * include-boundary / 2 emphasize-boundary.  Returns the number of filters, the width and spacing the boundary ends up with and up to
 * `cap` centres.  PINNED on the reference's function text (oracle/ref/extract_fn.py filter_build: ref_filter_boundary). */
double orc_filter_center(int type, size_t i, double width, double spacing, double ncp, double fmin) {
    if (type == 0)
        return ORC_FMA(ncp, width, ORC_FMA(spacing, (double)i, fmin)); /* StretchToCover::center: two vfmadd in the default build */
    if (type == 1)
        return spacing * (double)(i + 1);
    return spacing * (double)i;
}

int orc_filter_boundary(int type, double width, double spacing, double ncp, double fmin, double fmaxw, double* width_out, double* spacing_out,
                        double* centers, int cap) {
    if (spacing == 0)
        spacing = ncp * width; /* Boundary::setSpacing */
    size_t n_filters;
    if (type == 0) {
        /* StretchToCover::getNumberOfFilters / init */
        n_filters       = (size_t)floor(orc_postprocess_nfilters((fmaxw - fmin - width) / spacing + 1));
        double coverage = ORC_FMA(spacing, (double)(n_filters - 1), width) / (fmaxw - fmin); /* vfmadd132sd */
        if (!(n_filters == 1 && coverage > 1 && !orc_almost_equal(coverage, 1))) {
            width /= coverage;
            spacing /= coverage;
        }
    }
    else if (type == 1) /* IncludeBoundary::getNumberOfFilters; inverseWarpingFunction_ is the identity */
        n_filters = (size_t)ceil(orc_postprocess_nfilters(ORC_FMA(-(1 - ncp), width, fmaxw) / spacing)); /* vfnmadd132sd */
    else /* EmphasizeBoundary::getNumberOfFilters */
        n_filters = (size_t)floor(orc_postprocess_nfilters(fmaxw / spacing + 1));
    *width_out   = width;
    *spacing_out = spacing;
    for (size_t i = 0; i < n_filters && (int)i < cap; ++i)
        centers[i] = orc_filter_center(type, i, width, spacing, ncp, fmin);
    return (int)n_filters;
}

static int orc_build_filterbank(orc_mfcc* h) {
    const orc_mfcc_cfg* c  = &h->cfg;
    const orc_warp mel = {orc_mel, orc_mel_derivative, orc_mel_inverse}, bark = {orc_bark, orc_bark_derivative, orc_bark_inverse};
    if (c->warping < 0 || c->warping > 1 || c->filter_type < 0 || c->filter_type > 1 || c->boundary < 0 || c->boundary > 2)
        return -1;
    const orc_warp* W = c->warping == 1 ? &bark : &mel;
    /* FilterBankNode::configure reads sample-rate = N/fs from the attribute text */
    double sr_attr = orc_attr_roundtrip((double)h->fft_len / c->sample_rate);
    double d2c     = 1 / sr_attr;                    /* createScaling(1 / sampleRate_) */
    int    B       = h->n_bins;
    double fmin    = 0.0;                            /* filtering-interval-start default */
    double fmaxw   = W->value(d2c * (double)(B - 1)); /* FilterBankNode::init */
    h->mel_max     = fmaxw;

    double width   = c->mel_filter_width;
    double spacing = c->mel_spacing;
    /* normalizedCenterPosition: symmetrical triangle 0.5, trapeze 2.5 / 3.8 */
    double ncp     = c->filter_type == 1 ? 2.5 / (1.3 - (-2.5)) : 0.5;
    int nf = orc_filter_boundary(c->boundary, width, spacing, ncp, fmin, fmaxw, &width, &spacing, NULL, 0);
    if (nf < 0)
        return -1;
    size_t n_filters = (size_t)nf;
    h->n_filters = (int)n_filters;
    h->f_start   = (int*)calloc(n_filters, sizeof(int));
    h->f_end     = (int*)calloc(n_filters, sizeof(int));
    h->f_off     = (int*)calloc(n_filters + 1, sizeof(int));
    h->f_weights = (float*)calloc(n_filters * (size_t)B, sizeof(float));
    int off      = 0;
    for (size_t i = 0; i < n_filters; ++i) {
        double center = orc_filter_center(c->boundary, i, width, spacing, ncp, fmin);
        int start, end;
        int n = orc_filter_build(c->filter_type, c->warping, center, width, fmin, fmaxw, d2c, c->warp_differential_unit, &start, &end,
                                 h->f_weights + off, B);
        if (n < 0 || end > B)
            return -1; /* (end > B: Filter::apply would read beyond the spectrum) */
        h->f_start[i] = start;
        h->f_end[i]   = end;
        h->f_off[i]   = off;
        off += n;
    }
    h->f_off[n_filters] = off;

    if (c->front_end == 2) {
        /* plp.flow: signal-vector-f32-continuous-transform f = nest(nest(disc-to-cont, invert(bark)), equal-loudness-preemphasis)
         * on the vector [first, filters..., last] (Signal/VectorTransform.cc:36-83; Math/AnalyticFunctionFactory.cc:161-180:
         * "nest(g, f)" is f o g and f is created with maximal argument g(max); :322-327 disc-to-cont = scaling(1 / sample-rate),
         * sample-rate = the filter bank's output attribute 1 / spacing (Boundary::outputSampleRate) as text; :543-556 the 4 kHz
         * variant unless the largest frequency is significantly greater than 4000) */
        if (c->boundary == 0)
            return -1; /* stretch-to-cover reports sample rate 1: not the bark axis */
        size_t n_in = n_filters + 2;
        double sr   = orc_attr_roundtrip((double)1 / spacing);
        double g    = 1 / sr;
        double top  = orc_bark_inverse(g * (double)(n_in - 1));
        int    full = top > 4000.0 && !orc_almost_equal_tol(top, 4000.0, 1e12);
        h->eql      = (double*)calloc(n_in, sizeof(double));
        for (size_t i = 0; i < n_in; ++i) {
            double f  = orc_bark_inverse(g * (double)i);
            h->eql[i] = full ? orc_equal_loudness(f) : orc_equal_loudness_4khz(f);
        }
    }
    return 0;
}

/* Signal/CosineTransform.cc:62-74 (even about N - 1/2) and :46-60 (N-plus-one input data: the inverse DFT of an even spectrum sampled
 * at N + 1 points, which turns the compressed mel spectrum into autocorrelation coefficients), identity warping (the derivative is the
 * constant 1: `* 1.0`).  table [rows x cols] f32.  PINNED on the reference's function text (oracle/ref/extract_fn.py cosine_transform). */
void orc_cosine_table(int n_plus_one, int rows, int cols, float* table) {
    if (n_plus_one) {
        size_t N = (size_t)cols - 1;
        for (size_t k = 0; k < (size_t)rows; ++k) {
            table[k * cols + 0] = (float)0.5;
            table[k * cols + N] = (float)(0.5 * pow(-1, (double)k));
            for (size_t n = 1; n < N; ++n) {
                double omega        = M_PI * n / N;
                table[k * cols + n] = (float)(cos(omega * k) * 1.0);
            }
        }
    }
    else {
        size_t N = (size_t)cols;
        for (size_t k = 0; k < (size_t)rows; ++k)
            for (size_t n = 0; n < N; ++n) {
                double omega     = M_PI * (n + 0.5) / N;
                table[k * N + n] = (float)(cos(omega * k) * 1.0);
            }
    }
}

/* CosineTransform::init + apply (Signal/CosineTransform.cc:24-44,76-83): out = T * in with Math::Vector's left-to-right f32 dot product
 * (one fused multiply-add per term in the default build), divided by N_ (f32) when `normalize` -- N_ = cols for even-about-N-minus-half,
 * cols - 1 for N-plus-one */
void orc_cosine_transform(int n_plus_one, int n_in, int n_out, int normalize, const float* in, float* out, float* table_out) {
    float* T = (float*)calloc((size_t)n_out * n_in, sizeof(float));
    orc_cosine_table(n_plus_one, n_out, n_in, T);
    for (int k = 0; k < n_out; ++k) {
        float acc = 0;
        for (int n = 0; n < n_in; ++n)
            acc = ORC_FMAF(T[(size_t)k * n_in + n], in[n], acc);
        if (normalize)
            acc = acc / (float)(n_plus_one ? n_in - 1 : n_in);
        out[k] = acc;
    }
    if (table_out)
        memcpy(table_out, T, (size_t)n_out * n_in * sizeof(float));
    free(T);
}

static void orc_build_dct(orc_mfcc* h) {
    h->dct = (float*)calloc((size_t)h->n_ceps * (size_t)h->n_filters, sizeof(float));
    orc_cosine_table(0, h->n_ceps, h->n_filters, h->dct);
}

static void orc_build_cosine_nplus1(orc_mfcc* h) {
    size_t cols = (size_t)h->n_filters + (h->cfg.front_end == 2 ? 2 : 0), rows = (size_t)h->cfg.n_autocorrelation;
    h->dct      = (float*)calloc(rows * cols, sizeof(float));
    orc_cosine_table(1, (int)rows, (int)cols, h->dct);
}

/* Math::LevinsonLeastSquares (Math/LevinsonLse.cc:35-70): f64 recursion on f32 autocorrelation values; note the f32
 * division in the first reflection coefficient (-R[1] / R[0] is evaluated before it is widened) */
static int orc_almost_zero(double e) { /* Core::isAlmostEqual(e, 0.0), Core/Utility.hh:322-327 */
    double d = fabs(e);
    double t = (fabs(e) + 0.0 + 2.2250738585072014e-308) * 2.2204460492503131e-16 * 1.0;
    return d < t;
}

int orc_levinson(const float* R, int n, float* gain, float* a) {
    int N = n - 1;
    if (N < 1)
        return 0;
    double E[N + 1], k[N + 1], al[N + 1][N + 1];
    memset(al, 0, sizeof al);
    E[0] = R[0];
    if (orc_almost_zero(E[0]))
        return 0;
    al[1][1] = k[1] = -R[1] / R[0];
    E[1]            = R[0] + R[1] * k[1];
    for (int i = 2; i <= N; ++i) {
        k[i] = R[i];
        for (int j = 1; j <= i - 1; ++j)
            k[i] += al[j][i - 1] * R[i - j];
        if (orc_almost_zero(E[i - 1]))
            return 0;
        k[i]     = -k[i] / E[i - 1];
        al[i][i] = k[i];
        for (int j = 1; j <= i - 1; ++j)
            al[j][i] = al[j][i - 1] + k[i] * al[i - j][i - 1];
        E[i] = (1.0 - k[i] * k[i]) * E[i - 1];
    }
    *gain = (float)sqrt(E[N]);           /* gain(): sqrt(predictionError()), stored in AutoregressiveCoefficients::gain_ (f32) */
    for (int j = 1; j <= N; ++j)
        a[j - 1] = (float)al[j][N];
    return 1;
}

/* Signal/AutoregressionToCepstrum.cc:21-35 (PINNED on the function text in both builds).  log(gain) resolves to the double overload in that translation unit (only <cmath>
 * is in its include closure), integer factors are converted to f32, products run left to right in f32. */
void orc_ar_to_cepstrum(float gain, const float* a, int na, float* c, int nc) {
    (void)na;
    c[0] = (float)(2 * log((double)gain));
    c[1] = -a[0];
    for (int n = 2; n < nc; ++n) {
        c[n] = (float)n * a[n - 1];
        for (int k = 1; k < n; ++k) {
            float t = (float)(n - k) * c[n - k];
            c[n]    = ORC_FMAF(t, a[k - 1], c[n]); /* c[n] += (n - k) * c[n - k] * a[k - 1]: the second product is fused in the default build
                                                      (vfmadd132ss; function-text pin ar_to_cepstrum) */
        }
        c[n] = c[n] / (-(float)n);
    }
}

/* Signal/FastFourierTransform.cc:30-41 and FastFourierTransform.hh:299-308 */
static int orc_fft_length(double max_input_s, double fs) {
    unsigned maxlen = (unsigned)ceil(max_input_s * fs);
    if (maxlen == 0)
        return 0;
    double power = log((double)maxlen) / log((double)2);
    /* Core::isAlmostEqual(power, rint(power)) */
    if (orc_almost_equal(power, rint(power)))
        power = rint(power);
    else
        power = ceil(power);
    return 1 << (unsigned)power;
}

orc_mfcc* orc_mfcc_create(const orc_mfcc_cfg* cfg) {
    orc_mfcc* h = (orc_mfcc*)calloc(1, sizeof *h);
    h->cfg      = *cfg;
    /* Signal/Window.cc:69-80: rint of seconds * sample rate */
    h->frame_len   = (int)(unsigned)rint(cfg->win_len_s * cfg->sample_rate);
    h->frame_shift = (int)(unsigned)rint(cfg->win_shift_s * cfg->sample_rate);
    h->fft_len     = orc_fft_length(cfg->fft_max_input_s, cfg->sample_rate);
    if (h->frame_len <= 0 || h->frame_shift <= 0 || h->fft_len < h->frame_len) {
        free(h);
        return NULL;
    }
    h->n_bins    = h->fft_len / 2 + 1;
    h->n_ceps    = cfg->n_ceps;
    h->fft_scale = 1 / (float)cfg->sample_rate; /* Signal/FastFourierTransform.cc:66-73 */
    h->window    = (float*)calloc((size_t)h->frame_len, sizeof(float));
    orc_build_hamming(h->window, h->frame_len);
    if (orc_build_filterbank(h) != 0) {
        orc_mfcc_destroy(h);
        return NULL;
    }
    if (cfg->front_end == 1 || cfg->front_end == 2) {
        /* CosineTransformNode: nr-outputs <= input size; AutoregressionToCepstrumNode::init: 2 <= nr-outputs <= order + 1 */
        if (cfg->n_autocorrelation < 2 || cfg->n_autocorrelation > h->n_filters + (cfg->front_end == 2 ? 2 : 0) || cfg->n_ceps < 2 ||
            cfg->n_ceps > cfg->n_autocorrelation) {
            orc_mfcc_destroy(h);
            return NULL;
        }
        orc_build_cosine_nplus1(h);
    }
    else
        orc_build_dct(h);
    return h;
}

void orc_mfcc_destroy(orc_mfcc* h) {
    if (!h)
        return;
    free(h->window);
    free(h->f_start);
    free(h->f_end);
    free(h->f_off);
    free(h->f_weights);
    free(h->dct);
    free(h->eql);
    free(h);
}

int          orc_mfcc_frame_len(const orc_mfcc* h) { return h->frame_len; }
int          orc_mfcc_frame_shift(const orc_mfcc* h) { return h->frame_shift; }
int          orc_mfcc_fft_len(const orc_mfcc* h) { return h->fft_len; }
int          orc_mfcc_n_bins(const orc_mfcc* h) { return h->n_bins; }
int          orc_mfcc_n_filters(const orc_mfcc* h) { return h->n_filters; }
int          orc_mfcc_n_ceps(const orc_mfcc* h) { return h->n_ceps; }
const float* orc_mfcc_window(const orc_mfcc* h) { return h->window; }
const int*   orc_mfcc_filter_start(const orc_mfcc* h) { return h->f_start; }
const int*   orc_mfcc_filter_end(const orc_mfcc* h) { return h->f_end; }
const int*   orc_mfcc_filter_offset(const orc_mfcc* h) { return h->f_off; }
const float* orc_mfcc_filter_weights(const orc_mfcc* h) { return h->f_weights; }
const float* orc_mfcc_dct(const orc_mfcc* h) { return h->dct; }
double       orc_mfcc_mel_max(const orc_mfcc* h) { return h->mel_max; }
const double* orc_mfcc_equal_loudness(const orc_mfcc* h) { return h->eql; }

/* Signal/WindowBuffer.cc:84-125 + Signal/SlidingAlgorithmNode.hh:60-79: get() emits full
 * frames while >= 2*max(len,shift) samples are buffered; at end of segment flush() keeps
 * emitting every `shift` samples until the remainder fits one window; the last frame is
 * short (not re-centred).  Closed form: */
long orc_mfcc_n_frames(const orc_mfcc* h, long n) {
    if (n <= 0)
        return 0;
    long L = h->frame_len > h->frame_shift ? h->frame_len : h->frame_shift;
    if (n <= L)
        return 1;
    return (n - L + h->frame_shift - 1) / h->frame_shift + 1;
}

/* FilterBank::Filter::apply (Signal/Filterbank.cc:65-71): `result += in[f] * weights_[f - start_]`, f32, ascending bin -- one
 * vfmadd231ss in the native build; pinned in both flavours by the function-text pin ref_filter_apply (oracle/ref/extract_fn.py) */
float orc_filter_apply(const float* in, int start, int end, const float* weights) {
    float acc = 0;
    for (int b = start; b < end; ++b)
        acc = ORC_FMAF(in[b], weights[b - start], acc);
    return acc;
}

static void orc_frame(const orc_mfcc* h, const float* pre, long n_samples, long frame,
                      float* windowed, float* spectrum, float* amplitude, float* mel,
                      float* logmel, float* ceps) {
    const int N = h->fft_len;
    float     buf[N + 2];
    long      start = frame * (long)h->frame_shift;
    long      avail = n_samples - start;
    int       len   = avail < h->frame_len ? (int)avail : h->frame_len;
    /* Window::transform -> WindowFunction::work (Signal/WindowFunction.hh:81-96) */
    for (int i = 0; i < len; ++i)
        buf[i] = h->window[i] * pre[start + i];
    /* FastFourierTransform::zeroPadding */
    for (int i = len; i < N; ++i)
        buf[i] = 0;
    if (windowed)
        memcpy(windowed, buf, (size_t)N * sizeof(float));
    orc_fft_real(buf, N);
    /* RealFastFourierTransform::unpack (Signal/FastFourierTransform.cc:88-94) */
    buf[N]     = buf[1];
    buf[N + 1] = 0;
    buf[1]     = 0;
    /* estimateContinuous (Signal/FastFourierTransform.cc:66-73) */
    if (h->cfg.apply_scale && h->cfg.sample_rate != 1)
        for (int i = 0; i < N + 2; ++i)
            buf[i] = buf[i] * h->fft_scale;
    if (spectrum)
        memcpy(spectrum, buf, (size_t)(N + 2) * sizeof(float));
    /* amplitude: std::abs(std::complex<f32>) (Signal/ComplexVectorFunction.hh:30-47) */
    float amp[h->n_bins];
    for (int k = 0; k < h->n_bins; ++k)
        amp[k] = hypotf(buf[2 * k], buf[2 * k + 1]);
    if (amplitude)
        memcpy(amplitude, amp, (size_t)h->n_bins * sizeof(float));
    /* mfplp.flow: generic-vector-f32-power value 2.  Flow::VectorPowerFunction<f32> (Flow/SimpleFunction.hh:143-153) calls the
     * UNQUALIFIED pow(v[i], parameter) on two floats; with the headers that file sees (<cmath>, no using-directive) that is ::pow(double,
     * double), narrowed to f32 on assignment -- checked with g++ against the reference's own headers (sizeof(pow(1.0f, 2.0f)) == 8) */
    if (h->cfg.front_end != 0)
        for (int k = 0; k < h->n_bins; ++k)
            amp[k] = (float)pow((double)amp[k], (double)2.0f);
    float fb[h->n_filters];
    for (int f = 0; f < h->n_filters; ++f)
        fb[f] = orc_filter_apply(amp, h->f_start[f], h->f_end[f], h->f_weights + h->f_off[f]);
    if (mel)
        memcpy(mel, fb, (size_t)h->n_filters * sizeof(float));
    if (h->cfg.front_end != 0) {
        int   n_in = h->n_filters;
        float ext[h->n_filters + 2];
        if (h->cfg.front_end == 2) {
            /* plp.flow: generic-vector-f32-split port 0 / reversed port 0 + generic-vector-f32-concat (Flow/VectorSplit.hh:104-135):
             * [fb[0], fb[0..n-1], fb[n-1]]; then in[i] = (f32)((f64)in[i] * f(i)) (Signal/VectorTransform.cc:78-83,
             * Math/SimpleAnalyticFunctions.hh MultiplicationFunction) */
            n_in   = h->n_filters + 2;
            ext[0] = fb[0];
            memcpy(ext + 1, fb, (size_t)h->n_filters * sizeof(float));
            ext[n_in - 1] = fb[h->n_filters - 1];
            for (int i = 0; i < n_in; ++i)
                ext[i] = (float)((double)ext[i] * h->eql[i]);
        }
        else
            memcpy(ext, fb, (size_t)n_in * sizeof(float));
        /* intensity-loudness-law, autocorrelation (CosineTransform::apply: f32 rows left to right, divided by N_ = inputs - 1),
         * autoregression, cepstrum; a frame whose recursion fails is reported as an error by the reference: NaN here */
        const float pw = (float)h->cfg.plp_power;
        for (int f = 0; f < n_in; ++f)
            ext[f] = (float)pow((double)ext[f], (double)pw);   /* the same node: double pow of the f32 parameter, narrowed */
        if (logmel)
            memcpy(logmel, ext, (size_t)n_in * sizeof(float));
        if (ceps) {
            const int nac = h->cfg.n_autocorrelation;
            float     R[nac], a[nac], gain = 0;
            for (int k = 0; k < nac; ++k) {
                float        acc = 0;
                const float* row = h->dct + (size_t)k * n_in;
                for (int n = 0; n < n_in; ++n)
                    acc = ORC_FMAF(row[n], ext[n], acc);
                if (h->cfg.dct_normalize)
                    acc = acc / (float)(n_in - 1);
                R[k] = acc;
            }
            if (orc_levinson(R, nac, &gain, a))
                orc_ar_to_cepstrum(gain, a, nac - 1, ceps, h->n_ceps);
            else
                for (int k = 0; k < h->n_ceps; ++k)
                    ceps[k] = NAN;
        }
        return;
    }
    /* Flow::VectorLogFunction<f32> (Flow/SimpleFunction.hh:40-49): log10f, no floor */
    for (int f = 0; f < h->n_filters; ++f)
        fb[f] = log10f(fb[f]);
    if (logmel)
        memcpy(logmel, fb, (size_t)h->n_filters * sizeof(float));
    /* CosineTransform::apply (Signal/CosineTransform.cc:76-83; Math/Vector.hh:95-101) */
    if (ceps) {
        for (int k = 0; k < h->n_ceps; ++k) {
            float        acc = 0;
            const float* row = h->dct + (size_t)k * h->n_filters;
            for (int n = 0; n < h->n_filters; ++n)
                acc = ORC_FMAF(row[n], fb[n], acc); /* Math::Vector::operator*: result += a[i] * b[i] -- one vfmadd231ss in the native build of ref_matrix_vector */
            if (h->cfg.dct_normalize)
                acc = acc / (float)h->n_filters;
            ceps[k] = acc;
        }
    }
}

long orc_mfcc_run(const orc_mfcc* h, const float* pcm, long n_samples, float* ceps) {
    long T = orc_mfcc_n_frames(h, n_samples);
    if (T == 0)
        return 0;
    float* pre = (float*)malloc((size_t)n_samples * sizeof(float));
    memcpy(pre, pcm, (size_t)n_samples * sizeof(float));
    orc_preemphasis(pre, n_samples, (float)h->cfg.preemph_alpha);
    for (long t = 0; t < T; ++t)
        orc_frame(h, pre, n_samples, t, NULL, NULL, NULL, NULL, NULL, ceps + t * h->n_ceps);
    free(pre);
    return T;
}

int orc_mfcc_stages(const orc_mfcc* h, const float* pcm, long n_samples, long frame,
                    float* windowed, float* spectrum, float* amplitude, float* mel,
                    float* logmel, float* ceps) {
    long T = orc_mfcc_n_frames(h, n_samples);
    if (frame < 0 || frame >= T)
        return -1;
    float* pre = (float*)malloc((size_t)n_samples * sizeof(float));
    memcpy(pre, pcm, (size_t)n_samples * sizeof(float));
    orc_preemphasis(pre, n_samples, (float)h->cfg.preemph_alpha);
    orc_frame(h, pre, n_samples, frame, windowed, spectrum, amplitude, mel, logmel, ceps);
    free(pre);
    return 0;
}

/* ------------------------------------------------------------------ signal-dc-detection (samples.flow)
 * Signal::DcDetection (Signal/DcDetection.cc:90-235, DcDetection.hh:75-84) kept literal: a sample buffer with the three counters,
 * put / get / flush driven like SlidingAlgorithmNode::work (get until it fails, then the next input block, flush at end of
 * stream).  `block` = size of the input vectors (the result must not depend on it).  Blocks are reported as (first sample, length).
 * PARITY UNPINNED: DcDetection.cc includes the Flow node headers (boost); nothing in the reference's tests exercises it. */
typedef struct {
    const float* x;     /* the whole segment; buffer_ = x[base .. have) */
    long long    base, have;
    unsigned     minDc, minSeg, maxOut;
    float        maxInc;
    unsigned long long nonDc, dc, segLen;
} orc_dcd;

static int orc_dcd_next_block(orc_dcd* s) {
    while ((long long)(s->nonDc + s->dc) < s->have - s->base) {
        float v = s->x[s->base + s->nonDc + s->dc], ref = s->x[s->base + s->nonDc - 1];
        if (fabs(v - ref) >= s->maxInc) { /* isNonDC */
            if (s->dc >= s->minDc)
                return 1;
            s->nonDc += s->dc;
            s->dc = 0;
            unsigned lim = s->minSeg > s->maxOut ? s->minSeg : s->maxOut;
            if (s->nonDc >= lim)
                return 1;
            s->nonDc++;
        }
        else
            s->dc++;
    }
    return 0;
}

static int orc_dcd_flush_block(orc_dcd* s, long long* starts, long long* lens, long long cap, long long* count) {
    int result = 0;
    if ((s->segLen += s->nonDc) >= s->minSeg) { /* copyBlock */
        if (*count < cap && starts) {
            starts[*count] = s->base;
            lens[*count]   = (long long)s->nonDc;
        }
        ++*count;
        result = 1;
    }
    if (s->dc > 0)
        s->segLen = 0;
    s->base += (long long)(s->nonDc + s->dc); /* eraseBlock */
    s->nonDc = 1;
    s->dc    = 0;
    return result;
}

long long orc_dc_detection(const float* pcm, long long n, long long block, double sample_rate, double min_dc_length_s, float max_dc_increment,
                           double min_non_dc_segment_length_s, int maximal_output_size, long long* starts, long long* lens, long long cap) {
    orc_dcd s = {pcm, 0, 0, (unsigned)rint(min_dc_length_s * sample_rate), (unsigned)rint(min_non_dc_segment_length_s * sample_rate),
                 (unsigned)maximal_output_size, max_dc_increment, 1, 0, 0};
    long long count = 0;
    int       eos = 0;
    if (block < 1)
        block = 1;
    for (;;) {
        /* get(): do { if (!nextBlock()) return false; } while (!flushBlock(out)); */
        int got = 0;
        for (;;) {
            if (!orc_dcd_next_block(&s))
                break;
            if (orc_dcd_flush_block(&s, starts, lens, cap, &count)) {
                got = 1;
                break;
            }
        }
        if (got)
            continue;
        if (s.have < n) { /* put the next input vector */
            s.have = s.have + block < n ? s.have + block : n;
            continue;
        }
        if (eos)
            break;
        /* flush(): lastBlock + flushBlock, once per stream */
        if (s.have - s.base > 0) {
            if (s.dc < s.minDc) {
                s.nonDc += s.dc;
                s.dc = 0;
            }
            orc_dcd_flush_block(&s, starts, lens, cap, &count);
        }
        eos = 1;
        if (s.have - s.base <= 0)
            break;
    }
    return count;
}
