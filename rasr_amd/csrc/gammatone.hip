// gammatone.hip -- RASR's gammatone front-end nodes on gfx950.
//
//   signal-gammatone            Signal/GammaTone.cc:20-231: centre frequencies on the Greenwood / ERB scale through a two-piece
//                               linear warping, ERB bandwidths, per channel a cascade of second-order sections
//                                   o -= b1 w1;  o -= b2 w2;  wn = o;  o *= a0;  o += a1 w1;  w2 = w1;  w1 = wn
//                               in f32, one rounding per operation, state carried through the segment
//   signal-temporalintegration  Signal/TemporalIntegration.cc:60-84 on Signal/TimeWindowBuffer.cc:52-125: frames of `length` every
//                               `shift` with WindowBuffer's flush rule (short last frames get a window of their own length);
//                               per channel  acc = o[0] w[0];  acc = (f32)((f64)acc + |o[i]| w[i])   (the reference's fabs is the
//                               double overload)
//   signal-spectralintegration  Signal/SpectralIntegration.cc:55-74: windowed sums over neighbouring channels
//   generic-vector-f32-power, signal-cosine-transform (optional tail)
//
// The recursion is sequential in time and must be evaluated in the reference's order to give its bits, so the parallel axes are
// channel (lane) and segment (workgroup): gammatone_filter_kernel runs one lane per (segment, channel) over the segment's samples --
// PCM is fetched 64 samples at a time, one per lane, and broadcast with v_readlane --, feeds every sample straight into the (at most
// 8) temporal-integration windows that contain it and writes one f32 per (frame, channel): the [samples x channels] filter output
// (43 MB per 10 s of audio at 68 channels) never exists unless a caller asks for it.  gammatone_post_kernel finishes the frame.
// The host side restates the node's coefficient design operation by operation (f32 members, double-overload libm calls,
// std::complex<f32> division and abs).
#include "common.hpp"

#include <cmath>
#include <cstring>
#include <vector>

struct amx_gammatone {
    amx_ctx*           ctx = nullptr;
    amx_gammatone_cfg  cfg;
    bool               fma = false;   // contract=fma (cfg.tuning, else the context's): the reference's default build
    int                channels = 0, cascade = 0, ti_len = 0, ti_shift = 0, si_channels = 0, n_out = 0;
    std::vector<float> cf, coef, ti_win, si_win, dct;
    float *            d_coef = nullptr, *d_ti_win = nullptr, *d_si_win = nullptr, *d_dct = nullptr;
    // per-call scratch
    float*     d_ti = nullptr;
    size_t     ti_cap = 0;
    long long* d_off = nullptr;  // [2][n_seg + 1] sample / frame offsets
    size_t     off_cap = 0;
    float *    d_pcm = nullptr, *d_out = nullptr;  // staging of the host entry point
    size_t     pcm_cap = 0, out_cap = 0;
};

namespace amx {

constexpr int kGtMaxCascade = 8;
constexpr int kGtMaxOverlap = 8;     // frames that can contain one sample: ceil(length / shift)
constexpr int kGtChunk      = 32;    // samples per producer / consumer hand-over (LDS: 2 x kGtChunk x 256 B per workgroup, which
                                     // decides how many (segment, 64 channels) workgroups a CU holds: 9 at 32, 4 at 64)
constexpr int kGtMaxWindow  = 4096;  // temporal-integration window kept in LDS (with the 16 KB chunk buffer: <= 32 KB per wave)

// not inlined: only the short frames at the end of a segment come here, and the f64 cosine would bloat the sample loop 16 times over
__device__ __noinline__ float gt_window(int type, int len, int i) {  // Signal/WindowFunction.cc:66-72,103-120, symmetric fill
    if (type == AMX_WINDOW_RECTANGULAR)
        return 1.f;
    if (len <= 1)
        return 0.f;  // a one-point window is never initialised by the reference; Hanning's first point is 0 in every history
    unsigned M = (unsigned)len - 1, n = (unsigned)i;
    if (n > M / 2)
        n = M - n;
    return (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / M));
}

struct GtParams {
    const float*     pcm;
    const long long* sample_off;  // [n_seg + 1]
    const long long* frame_off;   // [n_seg + 1]
    const float*     coef;        // [channels][4]
    const float*     ti_win;      // [ti_len]
    float*           ti;          // [total_frames][channels]
    float*           filtered;    // nullable [total_samples][channels]
    int              channels, cascade, ti_len, ti_shift, ti_window, overlap;
};

// CASCADE: compile-time cascade depth (0 = run-time value, bounded by kGtMaxCascade).
// A wave works through its segment in chunks of kGtChunk samples: the filter cascade runs over the chunk (one dependent f32 chain per lane)
// and leaves its outputs in LDS, then every temporal-integration slot -- frame f lives in slot f % overlap, at most one frame per
// slot at a time -- consumes the part of the chunk that belongs to its frame in a branch-free inner loop.  Frame openings and
// completions are wave-uniform events handled between those loops.
// The two halves of a chunk's work have nothing in common but the chunk itself, so they run as a producer / consumer pair of waves
// (on different SIMDs of the CU): wave 0 filters chunk k into one of two LDS buffers while wave 1 integrates chunk k - 1 from the
// other; one workgroup barrier per chunk.  The critical path per sample is the longer of the two instead of their sum.
// FMA: the reference's default build (the sites: the cascade's two vfnmadd132ss and one vfmadd132ss per
// section, the temporal integration's f64 vfmadd132sd).
template<int CASCADE, bool FMA>
__global__ __launch_bounds__(128) void gammatone_filter_kernel(GtParams p) {
    extern __shared__ float s_mem[];
    float*                  s_win = s_mem;                                 // [ti_len]
    float*                  s_ob  = s_mem + ((p.ti_len + 63) & ~63);       // [2][kGtChunk samples][64 lanes]
    const int               lane = threadIdx.x & 63;
    const int               role = threadIdx.x >> 6;                       // 0: filter cascade, 1: temporal integration
    const int               ch   = blockIdx.y * 64 + lane;
    const bool              live = ch < p.channels;
    for (int i = threadIdx.x; i < p.ti_len; i += 128)
        s_win[i] = p.ti_win[i];
    __syncthreads();
    const long long s0 = p.sample_off[blockIdx.x], f0 = p.frame_off[blockIdx.x];
    const int       Ni = (int)(p.sample_off[blockIdx.x + 1] - s0), Ti = (int)(p.frame_off[blockIdx.x + 1] - f0);
    const int       cc = live ? ch : p.channels - 1;
    const float a0 = p.coef[cc * 4 + 0], a1 = p.coef[cc * 4 + 1], b1 = p.coef[cc * 4 + 2], b2 = p.coef[cc * 4 + 3];
    constexpr int NC = CASCADE ? CASCADE : kGtMaxCascade;
    float         w1[NC], w2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
        w1[c] = w2[c] = 0.f;
    float acc[kGtMaxOverlap];
    int   left[kGtMaxOverlap], pos[kGtMaxOverlap], flen[kGtMaxOverlap], fidx[kGtMaxOverlap], nstart[kGtMaxOverlap], nframe[kGtMaxOverlap];
#pragma unroll
    for (int r = 0; r < kGtMaxOverlap; ++r) {
        acc[r]    = 0.f;
        left[r]   = 0;                // samples of the slot's frame still to come (0: slot idle)
        pos[r]    = 0;                // index inside the frame of the next sample
        flen[r]   = 0;
        fidx[r]   = 0;
        nframe[r] = r;                // the next frame this slot will hold, and the sample it starts at
        nstart[r] = r * p.ti_shift;
    }
    const bool hann = p.ti_window != AMX_WINDOW_RECTANGULAR;
    constexpr int CH = kGtChunk;
    const int n_chunks = (Ni + CH - 1) / CH;
    for (int kc = 0; kc <= n_chunks; ++kc) {
      if (role == 0 && kc < n_chunks) {
        const int   nb    = kc * CH;
        float*      s_o   = s_ob + (kc & 1) * (CH * 64);
        const float chunk = (lane < CH && nb + lane < Ni) ? p.pcm[s0 + nb + lane] : 0.f;
        const int   cnt   = Ni - nb < CH ? Ni - nb : CH;
        // ---- the cascade over the chunk
        for (int j = 0; j < cnt; ++j) {
            float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(chunk), j));
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (CASCADE || c < p.cascade) {
                    o              = amx::mad<FMA>(-b1, w1[c], o);   // out -= b1 * buffer0
                    o              = amx::mad<FMA>(-b2, w2[c], o);   // out -= b2 * buffer1
                    const float wn = o;
                    o              = o * a0;
                    o              = amx::mad<FMA>(a1, w1[c], o);    // out * a0 + a1 * buffer0: the SECOND product is the fused one
                    w2[c]          = w1[c];
                    w1[c]          = wn;
                }
            s_o[j * 64 + lane] = o;
            if (p.filtered && live)
                p.filtered[(s0 + nb + j) * p.channels + ch] = o;
        }
      }
      else if (role == 1 && kc > 0) {
        const int    nb  = (kc - 1) * CH;
        const float* s_o = s_ob + ((kc - 1) & 1) * (CH * 64);
        const int    cnt = Ni - nb < CH ? Ni - nb : CH;
        // ---- temporal integration of the previous chunk, slot by slot
#pragma unroll
        for (int r = 0; r < kGtMaxOverlap; ++r) {
            if (r >= p.overlap)
                break;
            int j = 0;
            while (j < cnt) {
                if (left[r] > 0) {
                    const int m = left[r] < cnt - j ? left[r] : cnt - j;
                    float     a = acc[r];
                    if (!hann || flen[r] == p.ti_len) {  // the usual case: window values from LDS
                        // eight outputs and eight window values are fetched before the eight dependent (f64 add, round to f32) steps:
                        // the LDS latency is paid once per block, not once per sample
                        const float* w  = s_win + pos[r];
                        const float* so = s_o + j * 64 + lane;
                        int          k  = 0;
                        for (; k + 8 <= m; k += 8) {
                            float ov[8], wv[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                ov[u] = so[(k + u) * 64];
                                wv[u] = hann ? w[k + u] : 1.f;
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                a = (float)amx::mad<FMA>(fabs((double)ov[u]), (double)wv[u], (double)a);
                        }
                        for (; k < m; ++k)
                            a = (float)amx::mad<FMA>(fabs((double)so[k * 64]), (double)(hann ? w[k] : 1.f), (double)a);
                    }
                    else
                        for (int k = 0; k < m; ++k)  // a short frame at the end of the segment: its own window
                            a = (float)amx::mad<FMA>(fabs((double)s_o[(j + k) * 64 + lane]), (double)gt_window(p.ti_window, flen[r], pos[r] + k), (double)a);
                    acc[r] = a;
                    j += m;
                    pos[r] += m;
                    left[r] -= m;
                    if (left[r] == 0 && live)
                        p.ti[(f0 + fidx[r]) * p.channels + ch] = a;
                }
                else if (nframe[r] < Ti && nstart[r] < nb + cnt) {  // the slot's next frame opens inside this chunk: o[0] * w[0]
                    j              = nstart[r] - nb;
                    const int rest = Ni - nstart[r];
                    flen[r]        = rest < p.ti_len ? rest : p.ti_len;
                    fidx[r]        = nframe[r];
                    const float w  = hann ? (flen[r] == p.ti_len ? s_win[0] : gt_window(p.ti_window, flen[r], 0)) : 1.f;
                    acc[r]         = s_o[j * 64 + lane] * w;
                    pos[r]         = 1;
                    left[r]        = flen[r] - 1;
                    if (left[r] == 0 && live)
                        p.ti[(f0 + fidx[r]) * p.channels + ch] = acc[r];
                    ++j;
                    nframe[r] += p.overlap;
                    nstart[r] += p.overlap * p.ti_shift;
                }
                else
                    break;
            }
        }
      }
      __syncthreads();  // chunk kc is complete in its buffer; the other buffer is free again
    }
}

struct GtPostParams {
    const float* ti;      // [frames][channels]
    const float* si_win;  // [si_length]
    const float* dct;     // [n_ceps][si_channels]
    float*       out;     // [frames][n_out]
    long long    frames;
    int          channels, si_length, si_shift, si_channels, n_ceps, dct_normalize, n_out, fma;
    float        power;
};

// one workgroup per frame: spectral integration (+ root compression) into LDS, then the cosine transform rows
__global__ __launch_bounds__(128) void gammatone_post_kernel(GtPostParams p) {
    extern __shared__ float s_si[];
    const long long         t = blockIdx.x;
    const float*            in = p.ti + t * p.channels;
    for (int ch = threadIdx.x; ch < p.si_channels; ch += 128) {
        float v;
        if (p.si_length > 0) {
            float acc = 0.f;
            for (int w = 0; w < p.si_length; ++w) {
                acc = p.fma ? __builtin_fmaf(p.si_win[w], in[ch * p.si_shift + w], acc) : acc + p.si_win[w] * in[ch * p.si_shift + w];   // out += w[k] * in[..]: vfmadd132ss in the default build
            }
            v = acc;
        }
        else
            v = in[ch];
        if (p.power != 0.f)
            v = (float)pow((double)v, (double)p.power);  // generic-vector-f32-power: unqualified pow on floats = ::pow(double, double), narrowed
        s_si[ch] = v;
        if (p.n_ceps == 0)
            p.out[t * p.n_out + ch] = v;
    }
    if (p.n_ceps == 0)
        return;
    __syncthreads();
    for (int k = threadIdx.x; k < p.n_ceps; k += 128) {
        const float* row = p.dct + (size_t)k * p.si_channels;
        float        acc = 0.f;
        for (int n = 0; n < p.si_channels; ++n) {
            acc = p.fma ? __builtin_fmaf(row[n], s_si[n], acc) : acc + row[n] * s_si[n];   // Math::Vector's dot product
        }
        if (p.dct_normalize)
            acc = acc / (float)p.si_channels;
        p.out[t * p.n_out + k] = acc;
    }
}

namespace {

struct GtWarp {  // Signal::WarpingFunction, all f32
    float factor, brk, maxf, beta = 0, b = 0, wbrk = 0;
    bool  check() const { return !(brk - maxf == 0) && !(factor <= 0) && !(factor * brk >= maxf); }
    void  init(bool fma) {
        beta = (fma ? std::fma(factor, brk, -maxf) : factor * brk - maxf) / (brk - maxf);   // vfmsub132ss in the default build
        b    = maxf * (1 - beta);
        wbrk = fma ? std::fma(beta, brk, b) : beta * brk + b;  // warping(freqBreak_): vfmadd
    }
    float inverse(float f) const { return f < wbrk ? f / factor : (f - b) / beta; }
};

float host_window(int type, int len, int i) {
    if (type == AMX_WINDOW_RECTANGULAR)
        return 1.f;
    if (len <= 1)
        return 0.f;
    unsigned M = (unsigned)len - 1, n = (unsigned)i;
    if (n > M / 2)
        n = M - n;
    return (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / M));
}

template<class T>
int gt_upload(T** dst, const std::vector<T>& src) {
    AMX_HIP(hipMalloc((void**)dst, std::max<size_t>(src.size(), 1) * sizeof(T)));
    if (!src.empty())
        AMX_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return AMX_OK;
}

}  // namespace
}  // namespace amx

extern "C" {

void amx_gammatone_default_cfg(amx_gammatone_cfg* c) {
    if (!c)
        return;
    c->sample_rate     = 16000.0;
    c->cascade         = 4;  // GammaToneNode's parameter defaults (Signal/GammaTone.cc:222-231)
    c->min_freq        = 100;
    c->max_freq        = 6000;
    c->q               = 9.264491981582191;
    c->channels        = 50;
    c->cf_mode         = AMX_GAMMATONE_HUMAN;
    c->warp_freq_break = 6600;
    c->warping_factor  = 1;
    c->ti_window       = AMX_WINDOW_HANNING;
    c->ti_length_s     = 0.025;
    c->ti_shift_s      = 0.01;
    c->si_window       = AMX_WINDOW_HANNING;
    c->si_length       = 0;
    c->si_shift        = 1;
    c->power           = 0;
    c->n_ceps          = 0;
    c->dct_normalize   = 0;
    c->tuning          = nullptr;
}

int amx_gammatone_create(amx_ctx* ctx, const amx_gammatone_cfg* c, amx_gammatone** out) {
    using namespace amx;
    AMX_REQUIRE(c && out, AMX_ERR_INVALID, "amx_gammatone_create: NULL argument");
    *out = nullptr;
    AMX_REQUIRE(c->sample_rate > 0, AMX_ERR_INVALID, "gammatone: sample rate (%f) is not positive", c->sample_rate);
    AMX_REQUIRE(c->channels >= 2 && c->channels <= 4096, AMX_ERR_INVALID, "gammatone: channels (%d) must be in 2..4096", c->channels);
    AMX_REQUIRE(c->cascade >= 0 && c->cascade <= kGtMaxCascade, AMX_ERR_UNSUPPORTED, "gammatone: cascade (%d) must be in 0..%d", c->cascade,
                kGtMaxCascade);
    AMX_REQUIRE(c->cf_mode == AMX_GAMMATONE_HUMAN || c->cf_mode == AMX_GAMMATONE_ERB, AMX_ERR_INVALID, "gammatone: unknown cfmode %d", c->cf_mode);
    AMX_REQUIRE((c->ti_window == AMX_WINDOW_HANNING || c->ti_window == AMX_WINDOW_RECTANGULAR) &&
                        (c->si_window == AMX_WINDOW_HANNING || c->si_window == AMX_WINDOW_RECTANGULAR),
                AMX_ERR_UNSUPPORTED, "gammatone: window type must be hanning or rectangular");
    AMX_REQUIRE(c->ti_length_s > 0 && c->ti_shift_s > 0, AMX_ERR_INVALID, "gammatone: temporal integration length / shift must be positive");
    amx::Tuning tune;
    std::string t_contract;
    {
        static const char* const keys[] = {"contract", nullptr};
        static const char* const cons[] = {"off", "fma", nullptr};
        if (!tune.parse(c->tuning, keys, "amx_gammatone_create") ||
            !tune.get_word("contract", ctx && ctx->contract == AMX_CONTRACT_FMA ? "fma" : "off", cons, &t_contract, "amx_gammatone_create"))
            return AMX_ERR_INVALID;
    }
    amx_gammatone* h = new amx_gammatone;
    h->ctx           = ctx;
    h->cfg           = *c;
    h->cfg.tuning    = nullptr;   // the caller's string is not kept
    h->fma           = t_contract == "fma";
    const bool fma   = h->fma;
    h->channels      = c->channels;
    h->cascade       = c->cascade;
    // ---- GammaTone::init with the node's f32 members
    const float minFreq = (float)c->min_freq, maxFreq = (float)c->max_freq, l = 24.7f, q = (float)c->q;
    GtWarp      warp{(float)c->warping_factor, (float)c->warp_freq_break, (float)(c->sample_rate / 2)};
    if (!warp.check()) {  // GammaToneNode::init: error("Maybe there is a problem with the warping function.")
        delete h;
        amx::set_error("gammatone: Maybe there is a problem with the warping function.");
        return AMX_ERR_INVALID;
    }
    warp.init(fma);
    float g[3];
    if (c->cf_mode == AMX_GAMMATONE_HUMAN) {
        g[0] = 165.4;
        g[1] = 0.88;
        g[2] = 2.1;
    }
    else {
        g[2] = 1 / (q * std::log((double)10));
        g[1] = 1.0;
        g[0] = l / (g[1] * g[2] * std::log((double)10));
    }
    h->cf.resize(h->channels);
    h->coef.resize((size_t)h->channels * 4);
    const float xMin  = std::log10((double)(minFreq / g[0] + g[1])) / g[2];
    const float xMax  = std::log10((double)(maxFreq / g[0] + g[1])) / g[2];
    const float scale = (xMax - xMin) / float((unsigned)(h->channels - 1));
    for (unsigned i = 0; i < (unsigned)h->channels; i++) {
        const float exponent = g[2] * (fma ? std::fma((float)i, scale, xMin) : xMin + i * scale);   // vfmadd132ss in the default build
        h->cf[i]             = warp.inverse((float)(g[0] * (std::pow(10.0, (double)exponent) - g[1])));
    }
    const float k1Erb = l, k2Erb = 1 / (l * q);
    const float dt    = 1. / c->sample_rate;
    for (int f = 0; f < h->channels; f++) {
        const float bw    = k1Erb * (k2Erb * h->cf[f] + 1.0);
        const float theta = 2. * M_PI * h->cf[f] * dt;
        const float Phi   = 2. * M_PI * bw * dt;
        const float alpha = -std::exp((double)-Phi) * std::cos((double)theta);
        const float b1    = 2. * alpha;
        const float b2    = std::exp((double)(-2 * Phi));
        // std::complex<f32> arithmetic as g++ / libgcc evaluate it for the reference: complex + real touches the real part only;
        // the division is libgcc's __divsc3, which (libgcc >= 12) forms the quotient in double and rounds once -- written out here
        // because this file is compiled by clang, whose runtime library has a different __divsc3; abs = cabsf = hypotf
        const float  b1r = (float)(b1 * std::cos((double)theta)), b1i = (float)(-b1 * std::sin((double)theta));
        const float  b2r = (float)(b2 * std::cos((double)(2 * theta))), b2i = (float)(-b2 * std::sin((double)(2 * theta)));
        const float  alr = (float)(alpha * std::cos((double)theta)), ali = (float)(-alpha * std::sin((double)theta));
        const float  nr = (b1r + b2r) + 1.0f, ni = b1i + b2i, dr = alr + 1.0f, di = ali;
        const double aa = nr, bb = ni, cc = dr, dd = di, denom = (cc * cc) + (dd * dd);
        const float  qr = (float)(((aa * cc) + (bb * dd)) / denom), qi = (float)(((bb * cc) - (aa * dd)) / denom);
        const float  a0 = hypotf(qr, qi);
        const float a1 = alpha * a0;
        h->coef[f * 4 + 0] = a0;
        h->coef[f * 4 + 1] = a1;
        h->coef[f * 4 + 2] = b1;
        h->coef[f * 4 + 3] = b2;
    }
    // ---- TemporalIntegration::init
    h->ti_len   = (int)(unsigned)std::rint(c->ti_length_s * c->sample_rate);
    h->ti_shift = (int)(unsigned)std::rint(c->ti_shift_s * c->sample_rate);
    bool ok     = h->ti_len >= 1 && h->ti_shift >= 1;
    if (ok && (h->ti_len > kGtMaxWindow || (h->ti_len + h->ti_shift - 1) / h->ti_shift > kGtMaxOverlap)) {
        delete h;
        amx::set_error("gammatone: temporal integration window of %d samples every %d is not supported (<= %d samples, <= %d overlapping)",
                       h->ti_len, h->ti_shift, kGtMaxWindow, kGtMaxOverlap);
        return AMX_ERR_UNSUPPORTED;
    }
    h->si_channels = h->channels;
    if (ok && c->si_length > 0) {
        ok = c->si_shift >= 1 && c->si_length <= h->channels;
        if (ok) {
            h->si_channels = (h->channels - c->si_length) / c->si_shift + 1;
            h->si_win.resize(c->si_length);
            for (int i = 0; i < c->si_length; ++i)
                h->si_win[i] = host_window(c->si_window, c->si_length, i);
        }
    }
    h->n_out = h->si_channels;
    if (ok && c->n_ceps > 0) {
        ok = c->n_ceps <= h->si_channels;  // CosineTransformNode: nr-outputs <= input size
        if (ok) {
            const size_t N = (size_t)h->si_channels;
            h->dct.resize((size_t)c->n_ceps * N);
            for (size_t k = 0; k < (size_t)c->n_ceps; ++k)
                for (size_t n = 0; n < N; ++n) {
                    const double omega = M_PI * (n + 0.5) / N;
                    h->dct[k * N + n]  = (float)(std::cos(omega * k) * 1.0);
                }
            h->n_out = c->n_ceps;
        }
    }
    if (!ok) {
        delete h;
        amx::set_error("gammatone: inconsistent integration / cosine transform sizes");
        return AMX_ERR_INVALID;
    }
    h->ti_win.resize(h->ti_len);
    for (int i = 0; i < h->ti_len; ++i)
        h->ti_win[i] = host_window(c->ti_window, h->ti_len, i);
    if (ctx) {
        AMX_HIP(hipSetDevice(ctx->device));
        int r;
        if ((r = gt_upload(&h->d_coef, h->coef)) != AMX_OK || (r = gt_upload(&h->d_ti_win, h->ti_win)) != AMX_OK ||
            (r = gt_upload(&h->d_si_win, h->si_win)) != AMX_OK || (r = gt_upload(&h->d_dct, h->dct)) != AMX_OK) {
            amx_gammatone_destroy(h);
            return r;
        }
    }
    *out = h;
    return AMX_OK;
}

void amx_gammatone_destroy(amx_gammatone* h) {
    if (!h)
        return;
    if (h->ctx) {
        hipSetDevice(h->ctx->device);
        hipFree(h->d_coef);
        hipFree(h->d_ti_win);
        hipFree(h->d_si_win);
        hipFree(h->d_dct);
        hipFree(h->d_ti);
        hipFree(h->d_off);
        hipFree(h->d_pcm);
        hipFree(h->d_out);
    }
    delete h;
}

int amx_gammatone_describe(const amx_gammatone* h, amx_gammatone_info* info) {
    AMX_REQUIRE(h && info, AMX_ERR_INVALID, "amx_gammatone_describe: NULL argument");
    info->channels    = h->channels;
    info->cascade     = h->cascade;
    info->frame_len   = h->ti_len;
    info->frame_shift = h->ti_shift;
    info->si_channels = h->si_channels;
    info->n_out       = h->n_out;
    return AMX_OK;
}

long amx_gammatone_n_frames(const amx_gammatone* h, long n) {
    if (!h || n <= 0)
        return 0;
    const long reach = std::max(h->ti_len, h->ti_shift);  // TimeWindowBuffer::get / flush: WindowBuffer's rule
    if (n <= reach)
        return 1;
    return (n - reach + h->ti_shift - 1) / h->ti_shift + 1;
}

int amx_gammatone_tables(const amx_gammatone* h, float* center_freq, float* coefficients) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gammatone_tables: NULL handle");
    if (center_freq)
        memcpy(center_freq, h->cf.data(), h->cf.size() * 4);
    if (coefficients)
        memcpy(coefficients, h->coef.data(), h->coef.size() * 4);
    return AMX_OK;
}

int amx_gammatone_run_batch_dev(amx_gammatone* h, int n_seg, const long* sample_offsets, const float* pcm_dev, float* out_dev,
                                float* filtered_dev) {
    using namespace amx;
    AMX_REQUIRE(h && n_seg >= 0 && (n_seg == 0 || (sample_offsets && pcm_dev && out_dev)), AMX_ERR_INVALID,
                "amx_gammatone_run_batch_dev: bad argument");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_gammatone_run_batch_dev: host-only handle (created without a context)");
    if (n_seg == 0)
        return AMX_OK;
    AMX_HIP(hipSetDevice(h->ctx->device));
    std::vector<long long> off(2 * ((size_t)n_seg + 1));
    long long*             so = off.data();
    long long*             fo = off.data() + n_seg + 1;
    so[0] = sample_offsets[0];
    fo[0] = 0;
    for (int u = 0; u < n_seg; ++u) {
        const long len = sample_offsets[u + 1] - sample_offsets[u];
        AMX_REQUIRE(len >= 0 && len <= 0x7fffffffL, AMX_ERR_INVALID, "amx_gammatone_run_batch_dev: segment %d has invalid length %ld", u, len);
        so[u + 1] = sample_offsets[u + 1];
        fo[u + 1] = fo[u] + amx_gammatone_n_frames(h, len);
    }
    const long long frames = fo[n_seg];
    if (frames == 0)
        return AMX_OK;
    if (off.size() > h->off_cap) {
        hipFree(h->d_off);
        h->d_off   = nullptr;
        h->off_cap = 0;
        AMX_HIP(hipMalloc((void**)&h->d_off, off.size() * 8));
        h->off_cap = off.size();
    }
    AMX_HIP(hipMemcpyAsync(h->d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice, h->ctx->stream));
    AMX_HIP(hipStreamSynchronize(h->ctx->stream));  // `off` is a local
    const bool   tail = h->cfg.si_length > 0 || h->cfg.power != 0 || h->cfg.n_ceps > 0;
    const size_t need = (size_t)frames * h->channels;
    if (tail && need > h->ti_cap) {
        hipFree(h->d_ti);
        h->d_ti   = nullptr;
        h->ti_cap = 0;
        AMX_HIP(hipMalloc((void**)&h->d_ti, need * 4));
        h->ti_cap = need;
    }
    GtParams p;
    p.pcm        = pcm_dev;
    p.sample_off = h->d_off;
    p.frame_off  = h->d_off + n_seg + 1;
    p.coef       = h->d_coef;
    p.ti_win     = h->d_ti_win;
    p.ti         = tail ? h->d_ti : out_dev;
    p.filtered   = filtered_dev;
    p.channels   = h->channels;
    p.cascade    = h->cascade;
    p.ti_len     = h->ti_len;
    p.ti_shift   = h->ti_shift;
    p.ti_window  = h->cfg.ti_window;
    p.overlap    = (h->ti_len + h->ti_shift - 1) / h->ti_shift;
    {
        ScopedKernelTimer timer(h->ctx, "gammatone");
        const dim3   grid(n_seg, (h->channels + 63) / 64);
        const size_t lds = (size_t)(((h->ti_len + 63) & ~63) + 2 * kGtChunk * 64) * 4;
        if (p.cascade == 4)  // the node's default
            hipLaunchKernelGGL((h->fma ? gammatone_filter_kernel<4, true> : gammatone_filter_kernel<4, false>), grid, dim3(128), lds, h->ctx->stream, p);
        else
            hipLaunchKernelGGL((h->fma ? gammatone_filter_kernel<0, true> : gammatone_filter_kernel<0, false>), grid, dim3(128), lds, h->ctx->stream, p);
    }
    if (tail) {
        GtPostParams q;
        q.ti            = h->d_ti;
        q.si_win        = h->d_si_win;
        q.dct           = h->d_dct;
        q.out           = out_dev;
        q.frames        = frames;
        q.channels      = h->channels;
        q.si_length     = h->cfg.si_length;
        q.si_shift      = h->cfg.si_shift;
        q.si_channels   = h->si_channels;
        q.n_ceps        = h->cfg.n_ceps;
        q.dct_normalize = h->cfg.dct_normalize;
        q.n_out         = h->n_out;
        q.power         = (float)h->cfg.power;
        q.fma           = h->fma ? 1 : 0;
        hipLaunchKernelGGL(gammatone_post_kernel, dim3((unsigned)frames), dim3(128), (size_t)h->si_channels * 4, h->ctx->stream, q);
    }
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_gammatone_run(amx_gammatone* h, const float* pcm_host, long n_samples, float* out_host) {
    AMX_REQUIRE(h && n_samples >= 0 && (n_samples == 0 || (pcm_host && out_host)), AMX_ERR_INVALID, "amx_gammatone_run: bad argument");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_gammatone_run: host-only handle (created without a context)");
    const long T = amx_gammatone_n_frames(h, n_samples);
    if (T == 0)
        return AMX_OK;
    AMX_HIP(hipSetDevice(h->ctx->device));
    if ((size_t)n_samples > h->pcm_cap) {
        hipFree(h->d_pcm);
        h->d_pcm   = nullptr;
        h->pcm_cap = 0;
        AMX_HIP(hipMalloc((void**)&h->d_pcm, (size_t)n_samples * 4));
        h->pcm_cap = (size_t)n_samples;
    }
    const size_t on = (size_t)T * h->n_out;
    if (on > h->out_cap) {
        hipFree(h->d_out);
        h->d_out   = nullptr;
        h->out_cap = 0;
        AMX_HIP(hipMalloc((void**)&h->d_out, on * 4));
        h->out_cap = on;
    }
    AMX_HIP(hipMemcpyAsync(h->d_pcm, pcm_host, (size_t)n_samples * 4, hipMemcpyHostToDevice, h->ctx->stream));
    const long off[2] = {0, n_samples};
    const int  r      = amx_gammatone_run_batch_dev(h, 1, off, h->d_pcm, h->d_out, nullptr);
    if (r != AMX_OK)
        return r;
    AMX_HIP(hipMemcpyAsync(out_host, h->d_out, on * 4, hipMemcpyDeviceToHost, h->ctx->stream));
    AMX_HIP(hipStreamSynchronize(h->ctx->stream));
    return AMX_OK;
}

}  // extern "C"
