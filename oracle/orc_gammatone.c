/* oracle/orc_gammatone.c -- CPU restatement of RASR's gammatone front-end nodes (TEST INFRASTRUCTURE, see orc.h).
 *
 *   signal-gammatone            Signal/GammaTone.cc:20-231   (WarpingFunction, centre frequencies, ERB bandwidths, coefficients,
 *                                                            cascade of second-order sections with persistent state)
 *   signal-temporalintegration  Signal/TemporalIntegration.cc:60-84 on Signal/TimeWindowBuffer.cc:52-125 (the framing / flush rule
 *                                                            of WindowBuffer: short last frames, the window is re-made for them)
 *   signal-spectralintegration  Signal/SpectralIntegration.cc:55-74
 *   generic-vector-f32-power    Flow/SimpleFunction.hh:143-153 (unqualified pow on floats = ::pow(double, double), narrowed),
 *   signal-cosine-transform     Signal/CosineTransform.cc:62-83
 *   Hanning / rectangular       Signal/WindowFunction.cc:66-72,103-120
 *
 * signal-gammatone (WarpingFunction + GammaTone: filter design and cascade) is PINNED on the reference's function text in both builds
 * (oracle/ref/extract_fn.py gammatone, tests/test_contract.py), and so are the windows, TemporalIntegration (init, transform) and
 * SpectralIntegration::apply (extract_fn.py windows); the framing is pinned on Signal/TimeWindowBuffer.cc compiled unmodified, the
 * cosine transform on its own pin.  What remains read-only is the nodes' parameter handling.
 * Arithmetic types follow the source literally: the class members are f32, the unqualified exp / cos / sin / log10 / pow / log /
 * fabs calls resolve to the double overloads with the headers this translation unit sees (checked with g++ on the reference's
 * Flow/Vector.hh + Math/Complex.hh + Core/Utility.hh: sizeof(exp(1.0f)) == 8), std::complex<f32> division / abs are libgcc's
 * __divsc3 (written out, see orc_gammatone_create) and hypotf. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc.h"

struct orc_gammatone {
    orc_gammatone_cfg cfg;
    int    channels, cascade;
    float *cf, *bw;   /* [channels] */
    float* coef;      /* [channels][4] a0 a1 b1 b2 */
    int    ti_len, ti_shift;
    int    si_channels; /* channels after spectral integration (== channels when absent) */
    float* si_win;      /* [si_length] */
    int    n_out;       /* output dimension */
    float* dct;         /* [n_ceps][si_channels] */
};

/* WarpingFunction (Signal/GammaTone.cc:20-75): all members f32 */
typedef struct {
    float factor, brk, maxf, beta, b, wbrk;
} orc_gt_warp;

static int orc_gt_warp_check(const orc_gt_warp* w) {
    if (w->brk - w->maxf == 0)
        return 0;
    if (w->factor <= 0)
        return 0;
    if (w->factor * w->brk >= w->maxf)
        return 0;
    return 1;
}

static void orc_gt_warp_init(orc_gt_warp* w) {
    if (!orc_gt_warp_check(w)) {
        w->factor = 1.0f;
        w->brk    = 6600.0f;
        w->maxf   = 8000.0f;
    }
    w->beta = ORC_FMAF(w->factor, w->brk, -w->maxf) / (w->brk - w->maxf); /* vfmsub132ss in the default build */
    w->b    = w->maxf * (1 - w->beta);
    w->wbrk = ORC_FMAF(w->beta, w->brk, w->b); /* warping(freqBreak_): `f < freqBreak_` is false, so the upper branch; vfmadd */
}

static float orc_gt_inverse_warping(const orc_gt_warp* w, float f) {
    if (f < w->wbrk)
        return f / w->factor;
    return (f - w->b) / w->beta;
}

/* window functions of Signal/WindowFunction.cc, symmetric fill, f64 -> f32; value i of a window of `len` points.
 * A window of one point is never initialised by the reference (init() fails): Hanning's first point is 0 in every history. */
float orc_window_value(int type, int len, int i) { /* type 0 Hanning, 1 rectangular; PINNED on the function text (tests/test_contract.py) */
    if (type == 1) /* rectangular */
        return 1.0f;
    if (len <= 1)
        return 0.0f;
    unsigned M = (unsigned)len - 1, n = (unsigned)i;
    if (n > M / 2)
        n = M - n;
    return (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / M));
}

/* Signal::TemporalIntegration::transform (Signal/TemporalIntegration.cc:70-81) on ONE frame [rows x channels]: the first row times
 * w[0] in f32, then `out += fabs(x) * w[i]` -- fabs is the double overload there, so the product and the sum run in f64 and the result
 * is narrowed at every step (one vfmadd132sd in the default build).  window: 0 Hanning, 1 rectangular, of `rows` points.
 * PINNED on the reference's function text in both builds (oracle/ref/extract_fn.py windows). */
void orc_temporal_integrate(int window, const float* frame, int rows, int channels, float* out) {
    for (int ch = 0; ch < channels; ++ch) {
        float acc = frame[ch];
        acc       = acc * orc_window_value(window, rows, 0);
        for (int i = 1; i < rows; ++i)
            acc = (float)ORC_FMA(fabs((double)frame[(size_t)i * channels + ch]), (double)orc_window_value(window, rows, i), (double)acc);
        out[ch] = acc;
    }
}

/* Signal::SpectralIntegration::apply (Signal/SpectralIntegration.cc:58-75) on one row of `channels` values: `out += w[k] * in[ch * shift +
 * k]` in f32 (vfmadd132ss in the default build); win = the window table of `length` points.  Returns the number of outputs. */
int orc_spectral_integrate(const float* win, int length, int shift, const float* in, int channels, float* out) {
    int oc = (channels - length) / shift + 1;
    for (int ch = 0; ch < oc; ++ch) {
        float acc = 0;
        for (int w = 0; w < length; ++w)
            acc = ORC_FMAF(win[w], in[ch * shift + w], acc);
        out[ch] = acc;
    }
    return oc;
}

orc_gammatone* orc_gammatone_create(const orc_gammatone_cfg* c) {
    if (c->channels < 2 || c->cascade < 0 || c->sample_rate <= 0 || c->ti_length_s <= 0 || c->ti_shift_s <= 0 || c->ti_window < 0 ||
        c->ti_window > 1 || c->si_window < 0 || c->si_window > 1)
        return NULL;
    orc_gammatone* h = (orc_gammatone*)calloc(1, sizeof *h);
    h->cfg      = *c;
    h->channels = c->channels;
    h->cascade  = c->cascade;
    /* GammaToneNode: parameters are stored in f32 members; the warping function's maximum is sample-rate / 2 */
    const float minFreq = (float)c->min_freq, maxFreq = (float)c->max_freq, l = 24.7f, q = (float)c->q;
    orc_gt_warp w = {(float)c->warping_factor, (float)c->warp_freq_break, (float)(c->sample_rate / 2), 0, 0, 0};
    {   /* the node refuses a bad warping function ("Maybe there is a problem with the warping function.") */
        if (!orc_gt_warp_check(&w)) {
            free(h);
            return NULL;
        }
    }
    orc_gt_warp_init(&w);
    /* initializeCenterFrequencyList (:96-128) */
    float g[3];
    if (c->cf_mode == 0) {
        g[0] = 165.4;
        g[1] = 0.88;
        g[2] = 2.1;
    }
    else {
        g[2] = 1 / (q * log(10));
        g[1] = 1.0;
        g[0] = l / (g[1] * g[2] * log(10));
    }
    h->cf = (float*)calloc((size_t)h->channels, 4);
    h->bw = (float*)calloc((size_t)h->channels, 4);
    h->coef = (float*)calloc((size_t)h->channels * 4, 4);
    float xMin  = log10(minFreq / g[0] + g[1]) / g[2];
    float xMax  = log10(maxFreq / g[0] + g[1]) / g[2];
    float scale = (xMax - xMin) / (float)(unsigned)(h->channels - 1);
    for (unsigned i = 0; i < (unsigned)h->channels; i++) {
        float exponent = g[2] * ORC_FMAF((float)(unsigned)i, scale, xMin); /* xMin + i * scale: vfmadd132ss in the default build */
        h->cf[i]       = orc_gt_inverse_warping(&w, g[0] * (pow(10.0, exponent) - g[1]));
    }
    /* initBandWidths (:166-175) */
    float k1Erb = l, k2Erb = 1 / (l * q);
    for (int i = 0; i < h->channels; i++)
        h->bw[i] = k1Erb * (k2Erb * h->cf[i] + 1.0);
    /* initCoefficients (:133-161) */
    float dt = 1. / c->sample_rate;
    for (int f = 0; f < h->channels; f++) {
        float theta = 2. * M_PI * h->cf[f] * dt;
        float Phi   = 2. * M_PI * h->bw[f] * dt;
        float alpha = -exp(-Phi) * cos(theta);
        float b1    = 2. * alpha;
        float b2    = exp(-2 * Phi);
        /* std::complex<f32> b1C(b1 cos, -b1 sin), b2C(...), alphaC(...);  a0 = std::abs((b1C + b2C + 1.0f) / (alphaC + 1.0f)):
         * complex + real touches the real part only; the division is libgcc's __divsc3.  libgcc >= 12 forms the quotient in double
         * and rounds once (libgcc2.c, L_divsc3 with XMTYPE = double), older versions use Smith's method in float, which is one ulp
         * away in a0 on about a third of the channels -- which one a given RASR binary got depends on its toolchain (this very
         * container links gcc 11's static Smith version into a C shared object and gcc 12's libgcc_s into a g++ program).  The
         * newer formulation is written out here and in the product; abs = cabsf = hypotf. */
        float  b1r = (float)(b1 * cos(theta)), b1i = (float)(-b1 * sin(theta));
        float  b2r = (float)(b2 * cos(2 * theta)), b2i = (float)(-b2 * sin(2 * theta));
        float  alr = (float)(alpha * cos(theta)), ali = (float)(-alpha * sin(theta));
        float  nr = (b1r + b2r) + 1.0f, ni = b1i + b2i, dr = alr + 1.0f, di = ali;
        double aa = nr, bb = ni, cc = dr, dd = di, denom = (cc * cc) + (dd * dd);
        float  qr = (float)(((aa * cc) + (bb * dd)) / denom), qi = (float)(((bb * cc) - (aa * dd)) / denom);
        float  a0 = hypotf(qr, qi);
        float a1 = alpha * a0;
        h->coef[f * 4 + 0] = a0;
        h->coef[f * 4 + 1] = a1;
        h->coef[f * 4 + 2] = b1;
        h->coef[f * 4 + 3] = b2;
    }
    /* TemporalIntegration::init (:60-67) */
    h->ti_len   = (int)(unsigned)rint(c->ti_length_s * c->sample_rate);
    h->ti_shift = (int)(unsigned)rint(c->ti_shift_s * c->sample_rate);
    if (h->ti_len < 1 || h->ti_shift < 1) {
        orc_gammatone_destroy(h);
        return NULL;
    }
    h->si_channels = h->channels;
    if (c->si_length > 0) {
        if (c->si_shift < 1 || c->si_length > h->channels) {
            orc_gammatone_destroy(h);
            return NULL;
        }
        h->si_channels = (h->channels - c->si_length) / c->si_shift + 1;
        h->si_win      = (float*)calloc((size_t)c->si_length, 4);
        for (int i = 0; i < c->si_length; ++i)
            h->si_win[i] = orc_window_value(c->si_window, c->si_length, i);
    }
    h->n_out = h->si_channels;
    if (c->n_ceps > 0) {
        if (c->n_ceps > h->si_channels) {
            orc_gammatone_destroy(h);
            return NULL;
        }
        size_t N = (size_t)h->si_channels;
        h->dct   = (float*)calloc((size_t)c->n_ceps * N, 4);
        for (size_t k = 0; k < (size_t)c->n_ceps; ++k)
            for (size_t n = 0; n < N; ++n) {
                double omega      = M_PI * (n + 0.5) / N;
                h->dct[k * N + n] = (float)(cos(omega * k) * 1.0);
            }
        h->n_out = c->n_ceps;
    }
    return h;
}

void orc_gammatone_destroy(orc_gammatone* h) {
    if (!h)
        return;
    free(h->cf);
    free(h->bw);
    free(h->coef);
    free(h->si_win);
    free(h->dct);
    free(h);
}

int          orc_gammatone_n_out(const orc_gammatone* h) { return h->n_out; }
int          orc_gammatone_frame_len(const orc_gammatone* h) { return h->ti_len; }
int          orc_gammatone_frame_shift(const orc_gammatone* h) { return h->ti_shift; }
int          orc_gammatone_si_channels(const orc_gammatone* h) { return h->si_channels; }
const float* orc_gammatone_center_frequencies(const orc_gammatone* h) { return h->cf; }
const float* orc_gammatone_coefficients(const orc_gammatone* h) { return h->coef; }

/* TimeWindowBuffer::get / flush (Signal/TimeWindowBuffer.cc:83-125) driven by SlidingAlgorithmNode::work, flush-all = false:
 * get() delivers `length` samples and drops `shift` while the buffer holds >= 2 max(length, shift); at end of stream flush()
 * delivers min(length, rest) and stops once rest <= max(length, shift).  Frame f starts at f * shift.
 * PINNED on the reference class compiled unmodified (oracle/ref: ref_time_window_frames; tests/golden/ref_time_framing.json). */
long orc_time_window_frames(long n, int length, int shift, long* starts, int* lens, long cap) {
    if (n <= 0 || length <= 0 || shift <= 0)
        return 0;
    long L = length > shift ? length : shift, T = 0;
    for (long start = 0;; start += shift) {
        long rest = n - start;
        if (T < cap) {
            if (starts)
                starts[T] = start;
            if (lens)
                lens[T] = rest < length ? (int)rest : length;
        }
        ++T;
        if (rest <= L)
            break;
    }
    return T;
}

long orc_gammatone_n_frames(const orc_gammatone* h, long n) {
    return orc_time_window_frames(n, h->ti_len, h->ti_shift, 0, 0, 0);
}

/* filtered [n_samples x channels] (nullable), out [n_frames x n_out]; returns the number of frames */
long orc_gammatone_run(const orc_gammatone* h, const float* pcm, long n_samples, float* filtered, float* out) {
    const int C = h->channels, K = h->cascade;
    long      T = orc_gammatone_n_frames(h, n_samples);
    if (T == 0)
        return 0;
    /* GammaTone::apply (:197-218): state starts at zero with the segment (reset at eos) */
    float* y  = (float*)malloc((size_t)n_samples * C * 4);
    float* b0 = (float*)calloc((size_t)C * (K ? K : 1), 4);
    float* b1 = (float*)calloc((size_t)C * (K ? K : 1), 4);
    for (long i = 0; i < n_samples; ++i)
        for (int ch = 0; ch < C; ++ch) {
            const float* co = h->coef + ch * 4;
            float        o  = pcm[i];
            for (int c = 0; c < K; ++c) {
                /* the default build: two vfnmadd132ss, one vmulss, one vfmadd132ss (the SECOND product of out * a0 + a1 * buffer) */
                o        = ORC_FMAF(-co[2], b0[ch * K + c], o);
                o        = ORC_FMAF(-co[3], b1[ch * K + c], o);
                float wn = o;
                o        = o * co[0];
                o        = ORC_FMAF(co[1], b0[ch * K + c], o);
                b1[ch * K + c] = b0[ch * K + c];
                b0[ch * K + c] = wn;
            }
            y[i * C + ch] = o;
        }
    if (filtered)
        memcpy(filtered, y, (size_t)n_samples * C * 4);
    /* TemporalIntegration::transform (:69-80) per frame; SpectralIntegration::apply; power; cosine transform */
    float* ti = (float*)malloc((size_t)C * 4);
    float* si = (float*)malloc((size_t)h->si_channels * 4);
    for (long t = 0; t < T; ++t) {
        long start = t * (long)h->ti_shift;
        long avail = n_samples - start;
        int  len   = avail < h->ti_len ? (int)avail : h->ti_len;
        orc_temporal_integrate(h->cfg.ti_window, y + start * C, len, C, ti);
        if (h->cfg.si_length > 0)
            orc_spectral_integrate(h->si_win, h->cfg.si_length, h->cfg.si_shift, ti, C, si);
        else
            memcpy(si, ti, (size_t)C * 4);
        if (h->cfg.power != 0)
            for (int ch = 0; ch < h->si_channels; ++ch)
                si[ch] = (float)pow((double)si[ch], (double)(float)h->cfg.power);   /* ::pow(double, double), see the header */
        float* o = out + t * h->n_out;
        if (h->cfg.n_ceps > 0) {
            for (int k = 0; k < h->cfg.n_ceps; ++k) {
                float        acc = 0;
                const float* row = h->dct + (size_t)k * h->si_channels;
                for (int n = 0; n < h->si_channels; ++n)
                    acc = ORC_FMAF(row[n], si[n], acc); /* CosineTransform::apply: Math::Vector's dot product (pinned, orc_cosine_transform) */
                if (h->cfg.dct_normalize)
                    acc = acc / (float)h->si_channels;
                o[k] = acc;
            }
        }
        else
            memcpy(o, si, (size_t)h->si_channels * 4);
    }
    free(y); free(b0); free(b1); free(ti); free(si);
    return T;
}
