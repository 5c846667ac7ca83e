// mfcc.hip -- fused MFCC front-end for gfx950 and the amx_mfcc_* part of the C ABI.
//
// One kernel replaces the per-frame Flow chain of mfcc.flow (Tools/FeatureExtraction/share/
// mfcc.flow:8-34): preemphasis -> Hamming framing -> zero-pad -> real FFT (x 1/fs) -> |X| ->
// mel triangular filter bank -> log10 -> DCT-II.
//
// Mapping to the hardware
//   * a workgroup (4 wavefronts) owns a TILE of up to `frames_per_tile` consecutive frames of one
//     segment.  The PCM span of the tile ((FT-1)*shift + len + 1 samples) is read from HBM once,
//     coalesced, into LDS; the 2.5x frame overlap is served from LDS, not from memory.
//   * each 64-lane wavefront transforms one frame at a time: the fft_len real samples are
//     packed as NC = fft_len/2 complex points, 4 points per lane for NC = 256, and run through
//     an in-LDS Stockham radix-4 (+ one radix-2 stage when log2 NC is odd) with per-lane
//     twiddles held in registers across frames.  Only wavefront-level ordering is needed
//     between stages (DS operations of one wave complete in order), so the 4 waves of a
//     workgroup never wait for each other inside the frame loop.
//   * real split, 1/fs scale and amplitude are a lane-parallel epilogue of the FFT; the mel
//     filters (lane = filter) and the DCT (lane = cepstral coefficient) read LDS-resident tables.
//   * algorithmic HBM traffic per frame: shift*4 B of PCM in + n_ceps*4 B out (800 B at 16 kHz /
//     40 ceps) -- the kernel is bounded by HBM bandwidth once VALU/LDS time is below that.
//
// Numerics: f32 throughout like the reference; window/preemphasis/filter-bank/DCT use separate
// multiply and add (this TU is compiled with -ffp-contract=off) in the reference's summation
// order; the FFT butterflies use explicit fmaf with table twiddles, so spectra differ from the
// reference's f64-recurrence butterflies at the 1e-7 relative level ("parity").
#include "common.hpp"
#include "mfcc_tables.hpp"

#include <algorithm>
#include <cstring>

namespace amx {

struct MfccTile {
    long long sample_base;  // first sample of the segment in the concatenated PCM buffer
    long long out_frame;    // global index of the tile's first frame in the output
    int       n_samples;    // segment length
    int       frame0;       // first frame of the tile within the segment
    int       n_frames;     // frames in this tile
    int       pad_;
};

struct MfccParams {
    const float*    pcm;
    float*          ceps;
    const MfccTile* tiles;
    const float*    window;
    const int*      fstart;
    const int*      fend;
    const int*      foff;
    const float*    fweights;
    const float*    dct_t;  // transposed [n_filters][n_ceps]
    const float2*   tw;     // [NC]  e^{+2 pi i k / NC}
    const float2*   stw;    // [NC/2+1] e^{+pi i k / NC}
    int             frame_len, frame_shift, n_filters, n_ceps, n_weights;
    int             frames_per_tile;
    float           alpha, fft_scale;
    int             apply_scale, dct_normalize;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    // a * w with fused multiply-adds
    return make_float2(fmaf(a.x, w.x, -(a.y * w.y)), fmaf(a.x, w.y, a.y * w.x));
}

__device__ __forceinline__ void wave_sync() {
    // order this wave's LDS traffic for the compiler; the hardware keeps DS ops of a wave in order
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template<int NC>
struct FftPlan {
    static constexpr int log2nc() {
        int l = 0;
        for (int n = NC; n > 1; n >>= 1)
            ++l;
        return l;
    }
    static constexpr int L   = log2nc();
    static constexpr int S4  = L / 2;         // radix-4 stages
    static constexpr bool R2 = (L % 2) != 0;  // one trailing radix-2 stage
    static constexpr int B4  = (NC / 4 + 63) / 64;  // radix-4 butterflies per lane
    static constexpr int B2  = (NC / 2 + 63) / 64;  // radix-2 butterflies per lane
};

template<int NC>
__global__ __launch_bounds__(256) void mfcc_kernel(MfccParams p) {
    using P = FftPlan<NC>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    const MfccTile tile = p.tiles[blockIdx.x];
    const int      span = (tile.n_frames - 1) * p.frame_shift + p.frame_len;  // samples touched (clipped below)

    // ---- LDS carve-up
    const int span_max = (p.frames_per_tile - 1) * p.frame_shift + p.frame_len + 1;
    float*    s_pcm    = smem;                                   // [span_max] raw samples, [0] = predecessor
    float*    s_win    = s_pcm + ((span_max + 3) & ~3);          // [frame_len]
    float*    s_fw     = s_win + ((p.frame_len + 3) & ~3);       // [n_weights]
    float*    s_dct    = s_fw + ((p.n_weights + 3) & ~3);        // [n_filters * n_ceps] transposed
    int*      s_fs     = (int*)(s_dct + ((p.n_filters * p.n_ceps + 3) & ~3));
    int*      s_fe     = s_fs + p.n_filters;
    int*      s_fo     = s_fe + p.n_filters;
    float*    s_wave   = (float*)(s_fo + ((p.n_filters + 3) & ~3));
    const int wave_floats = 2 * NC + ((p.n_filters + 3) & ~3);
    float2*   s_z      = (float2*)(s_wave + wave * wave_floats);  // [NC] complex work buffer
    float*    s_amp    = (float*)s_z;                            // [NC+1] amplitudes (aliases s_z)
    float*    s_lm     = s_wave + wave * wave_floats + 2 * NC;   // [n_filters] log-mel

    // ---- stage the tile: PCM span (coalesced, once) and the small tables
    {
        const long long first = (long long)tile.frame0 * p.frame_shift;  // within segment
        const float*    src   = p.pcm + tile.sample_base;
        const int       avail = (int)min((long long)span, (long long)tile.n_samples - first);
        for (int i = tid; i <= avail; i += 256) {
            // s_pcm[i] = x[first + i - 1]; at the segment start the predecessor is x[0]
            long long g = first + i - 1;
            s_pcm[i]    = src[g < 0 ? 0 : g];
        }
        for (int i = tid; i < p.frame_len; i += 256)
            s_win[i] = p.window[i];
        for (int i = tid; i < p.n_weights; i += 256)
            s_fw[i] = p.fweights[i];
        for (int i = tid; i < p.n_filters * p.n_ceps; i += 256)
            s_dct[i] = p.dct_t[i];
        for (int i = tid; i < p.n_filters; i += 256) {
            s_fs[i] = p.fstart[i];
            s_fe[i] = p.fend[i];
            s_fo[i] = p.foff[i];
        }
    }

    // ---- per-lane twiddles, constant across frames
    float2 tw4[P::S4 > 1 ? P::S4 - 1 : 1][P::B4][3];
#pragma unroll
    for (int s = 1; s < P::S4; ++s) {
        const int Ns = 1 << (2 * s);
#pragma unroll
        for (int b = 0; b < P::B4; ++b) {
            const int j   = lane + 64 * b;
            const int k   = j & (Ns - 1);
            const int idx = k * (NC / (4 * Ns));
#pragma unroll
            for (int r = 0; r < 3; ++r)
                tw4[s - 1][b][r] = p.tw[((r + 1) * idx) & (NC - 1)];
        }
    }
    float2 tw2[P::B2];
    if (P::R2) {
        const int Ns = NC / 2;
#pragma unroll
        for (int b = 0; b < P::B2; ++b) {
            const int j = lane + 64 * b;
            tw2[b]      = p.tw[(j & (Ns - 1)) & (NC - 1)];  // k * NC/(2*Ns) = k
        }
    }
    __syncthreads();

    const float alpha   = p.alpha;
    const bool  alpha1  = (alpha == 1.0f);
    const float scale   = p.fft_scale;
    const bool  doscale = p.apply_scale != 0;

    for (int f = wave; f < tile.n_frames; f += 4) {
        const int       o     = f * p.frame_shift;  // offset of the frame in the tile span
        const long long start = ((long long)tile.frame0 + f) * p.frame_shift;
        const int       len   = (int)min((long long)p.frame_len, (long long)tile.n_samples - start);

        // windowed, pre-emphasised sample i of this frame (0 beyond the frame)
        auto sample = [&](int i) -> float {
            if (i >= len)
                return 0.f;
            float x    = s_pcm[o + i + 1];
            float prev = s_pcm[o + i];
            float y;
            if (alpha1)
                y = x - prev;  // Signal/Preemphasis.cc:69-75
            else {
                float prod = alpha * prev;  // :62-67, f32 product then f32 difference
                y          = x - prod;
            }
            return s_win[i] * y;  // WindowFunction::work
        };

        // ================= complex FFT of NC points, natural order in/out (Stockham)
        wave_sync();  // previous frame's readers of s_z / s_amp / s_lm are done
        if (P::S4 >= 1) {
            // first radix-4 stage (Ns = 1, twiddles are 1) is fed straight from the PCM tile
            float2 y[P::B4][4];
#pragma unroll
            for (int b = 0; b < P::B4; ++b) {
                const int j = lane + 64 * b;
                if (j < NC / 4) {
                    float2 x[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = j + r * (NC / 4);
                        x[r]        = make_float2(sample(2 * c), sample(2 * c + 1));
                    }
                    float2 a = make_float2(x[0].x + x[2].x, x[0].y + x[2].y);
                    float2 bb = make_float2(x[0].x - x[2].x, x[0].y - x[2].y);
                    float2 c = make_float2(x[1].x + x[3].x, x[1].y + x[3].y);
                    float2 d = make_float2(x[1].x - x[3].x, x[1].y - x[3].y);
                    y[b][0]  = make_float2(a.x + c.x, a.y + c.y);
                    y[b][1]  = make_float2(bb.x - d.y, bb.y + d.x);  // (x0-x2) + i(x1-x3)
                    y[b][2]  = make_float2(a.x - c.x, a.y - c.y);
                    y[b][3]  = make_float2(bb.x + d.y, bb.y - d.x);  // (x0-x2) - i(x1-x3)
                }
            }
#pragma unroll
            for (int b = 0; b < P::B4; ++b) {
                const int j = lane + 64 * b;
                if (j < NC / 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        s_z[4 * j + q] = y[b][q];
                }
            }
        }
        else {
            // NC == 2: no radix-4 stage; load the points
            for (int j = lane; j < NC; j += 64)
                s_z[j] = make_float2(sample(2 * j), sample(2 * j + 1));
        }
#pragma unroll
        for (int s = 1; s < P::S4; ++s) {
            const int Ns = 1 << (2 * s);
            wave_sync();
            float2 y[P::B4][4];
#pragma unroll
            for (int b = 0; b < P::B4; ++b) {
                const int j = lane + 64 * b;
                if (j < NC / 4) {
                    float2 x0 = s_z[j];
                    float2 x1 = cmul(s_z[j + NC / 4], tw4[s - 1][b][0]);
                    float2 x2 = cmul(s_z[j + 2 * (NC / 4)], tw4[s - 1][b][1]);
                    float2 x3 = cmul(s_z[j + 3 * (NC / 4)], tw4[s - 1][b][2]);
                    float2 a  = make_float2(x0.x + x2.x, x0.y + x2.y);
                    float2 bb = make_float2(x0.x - x2.x, x0.y - x2.y);
                    float2 c  = make_float2(x1.x + x3.x, x1.y + x3.y);
                    float2 d  = make_float2(x1.x - x3.x, x1.y - x3.y);
                    y[b][0]   = make_float2(a.x + c.x, a.y + c.y);
                    y[b][1]   = make_float2(bb.x - d.y, bb.y + d.x);
                    y[b][2]   = make_float2(a.x - c.x, a.y - c.y);
                    y[b][3]   = make_float2(bb.x + d.y, bb.y - d.x);
                }
            }
            wave_sync();
#pragma unroll
            for (int b = 0; b < P::B4; ++b) {
                const int j = lane + 64 * b;
                if (j < NC / 4) {
                    const int k  = j & (Ns - 1);
                    const int j0 = ((j - k) << 2) + k;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        s_z[j0 + q * Ns] = y[b][q];
                }
            }
        }
        if (P::R2) {
            const int Ns = NC / 2;
            wave_sync();
            float2 y[P::B2][2];
#pragma unroll
            for (int b = 0; b < P::B2; ++b) {
                const int j = lane + 64 * b;
                if (j < NC / 2) {
                    float2 x0 = s_z[j];
                    float2 x1 = cmul(s_z[j + NC / 2], tw2[b]);
                    y[b][0]   = make_float2(x0.x + x1.x, x0.y + x1.y);
                    y[b][1]   = make_float2(x0.x - x1.x, x0.y - x1.y);
                }
            }
            wave_sync();
#pragma unroll
            for (int b = 0; b < P::B2; ++b) {
                const int j = lane + 64 * b;
                if (j < NC / 2) {
                    // Ns == NC/2: j0 = j, outputs at j and j + Ns
                    s_z[j]      = y[b][0];
                    s_z[j + Ns] = y[b][1];
                }
            }
        }
        wave_sync();

        // ================= real split (Math/FastFourierTransform.cc:113-133), 1/fs, amplitude
        // pairs (i, NC-i), i = 1..NC/2-1; bins 0, NC/2 and NC (Nyquist) handled by lane 0
        constexpr int NPAIR = NC / 2 - 1;
        constexpr int PB    = (NPAIR + 63) / 64 > 0 ? (NPAIR + 63) / 64 : 1;
        float         amp_lo[PB], amp_hi[PB];
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int i = 1 + lane + 64 * b;
            if (i <= NPAIR) {
                const float2 za  = s_z[i];
                const float2 zb  = s_z[NC - i];
                const float2 w   = p.stw[i];
                const float  h1r = 0.5f * (za.x + zb.x);
                const float  h1i = 0.5f * (za.y - zb.y);
                const float  h2r = 0.5f * (za.y + zb.y);
                const float  h2i = -0.5f * (za.x - zb.x);
                float        ar  = fmaf(-w.y, h2i, fmaf(w.x, h2r, h1r));
                float        ai  = fmaf(w.y, h2r, fmaf(w.x, h2i, h1i));
                float        br  = fmaf(w.y, h2i, fmaf(-w.x, h2r, h1r));
                float        bi  = fmaf(w.y, h2r, fmaf(w.x, h2i, -h1i));
                if (doscale) {
                    ar *= scale;
                    ai *= scale;
                    br *= scale;
                    bi *= scale;
                }
                amp_lo[b] = sqrtf(fmaf(ar, ar, ai * ai));
                amp_hi[b] = sqrtf(fmaf(br, br, bi * bi));
            }
        }
        float amp0 = 0, ampn = 0, ampm = 0;
        if (lane == 0) {
            const float2 z0 = s_z[0];
            float        dc = z0.x + z0.y, ny = z0.x - z0.y;
            if (doscale) {
                dc *= scale;
                ny *= scale;
            }
            amp0 = fabsf(dc);
            ampn = fabsf(ny);
            if (NC >= 2) {
                float2 zm = s_z[NC / 2];  // untouched by the reference's split loop
                if (doscale) {
                    zm.x *= scale;
                    zm.y *= scale;
                }
                ampm = sqrtf(fmaf(zm.x, zm.x, zm.y * zm.y));
            }
        }
        wave_sync();
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int i = 1 + lane + 64 * b;
            if (i <= NPAIR) {
                s_amp[i]      = amp_lo[b];
                s_amp[NC - i] = amp_hi[b];
            }
        }
        if (lane == 0) {
            s_amp[0]      = amp0;
            s_amp[NC]     = ampn;
            s_amp[NC / 2] = ampm;
        }
        wave_sync();

        // ================= mel filter bank (Signal/Filterbank.cc:65-71): lane = filter,
        // f32 accumulate in ascending bin order; then log10 (Flow/SimpleFunction.hh:40-49)
        for (int flt = lane; flt < p.n_filters; flt += 64) {
            const int    b0  = s_fs[flt], b1 = s_fe[flt];
            const float* w   = s_fw + s_fo[flt] - b0;
            float        acc = 0.f;
            for (int b = b0; b < b1; ++b) {
                float prod = s_amp[b] * w[b];
                acc        = acc + prod;
            }
            s_lm[flt] = log10f(acc);
        }
        wave_sync();

        // ================= DCT-II (Signal/CosineTransform.cc:76-83): lane = coefficient
        float* out = p.ceps + (tile.out_frame + f) * (long long)p.n_ceps;
        for (int k = lane; k < p.n_ceps; k += 64) {
            float acc = 0.f;
            for (int n = 0; n < p.n_filters; ++n) {
                float prod = s_dct[n * p.n_ceps + k] * s_lm[n];
                acc        = acc + prod;
            }
            if (p.dct_normalize)
                acc = acc / (float)p.n_filters;
            out[k] = acc;
        }
    }
}

// out[t] = concat(x[clamp(t-left)], ..., x[clamp(t+right)]) per segment
__global__ __launch_bounds__(256) void context_window_kernel(const float* __restrict__ feats, const long long* __restrict__ frame_off,
                                                            int n_seg, int dim, int left, int right,
                                                            float* __restrict__ out, int out_stride, long long total) {
    const long long t = blockIdx.x;
    if (t >= total)
        return;
    // binary search of the segment containing frame t
    int lo = 0, hi = n_seg;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (frame_off[mid] <= t)
            lo = mid;
        else
            hi = mid;
    }
    const long long s0 = frame_off[lo], s1 = frame_off[lo + 1];
    const int       w  = (left + right + 1) * dim;
    for (int i = threadIdx.x; i < out_stride; i += blockDim.x) {
        float v = 0.f;
        if (i < w) {
            int       c  = i / dim, d = i - c * dim;
            long long tt = t - left + c;
            tt           = tt < s0 ? s0 : (tt >= s1 ? s1 - 1 : tt);
            v            = feats[tt * dim + d];
        }
        out[t * (long long)out_stride + i] = v;
    }
}

}  // namespace amx

// ------------------------------------------------------------------------------------ ABI

struct amx_mfcc {
    amx_ctx*        ctx = nullptr;
    amx::MfccTables tab;
    int             frames_per_tile = 32;
    // device copies of the tables
    float * d_window = nullptr, *d_fw = nullptr, *d_dct_t = nullptr;
    int *   d_fs = nullptr, *d_fe = nullptr, *d_fo = nullptr;
    float2 *d_tw = nullptr, *d_stw = nullptr;
    size_t  lds_bytes = 0;
};

struct amx_mfcc_plan {
    amx_mfcc*              owner = nullptr;
    int                    n_seg = 0;
    std::vector<long>      sample_off, frame_off;
    std::vector<amx::MfccTile> tiles;
    amx::MfccTile*         d_tiles     = nullptr;
    long long*             d_frame_off = nullptr;
};

namespace {

template<class T>
int upload(T** dst, const T* src, size_t n) {
    AMX_HIP(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
    if (n)
        AMX_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return AMX_OK;
}

size_t mfcc_lds_bytes(const amx::MfccTables& t, int ft) {
    auto   r4       = [](size_t v) { return (v + 3) & ~(size_t)3; };
    size_t span_max = (size_t)(ft - 1) * t.frame_shift + t.frame_len + 1;
    size_t fl       = r4(span_max) + r4(t.frame_len) + r4(t.filter_weights.size()) + r4((size_t)t.n_filters * t.n_ceps);
    fl += 2 * (size_t)t.n_filters + r4(t.n_filters);  // ints
    fl += 4 * ((size_t)t.fft_len + r4(t.n_filters));  // per wave: 2*NC floats + log-mel
    return fl * 4;
}

template<int NC>
int launch_mfcc(amx_mfcc* h, const amx::MfccParams& p, int n_tiles) {
    if (n_tiles <= 0)
        return AMX_OK;
    auto kern = amx::mfcc_kernel<NC>;
    if (h->lds_bytes > 64 * 1024)
        AMX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
    amx::ScopedKernelTimer timer(h->ctx, "mfcc");
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(256), h->lds_bytes, h->ctx->stream, p);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

}  // namespace

extern "C" {

void amx_mfcc_default_cfg(amx_mfcc_cfg* c) {
    if (!c)
        return;
    c->sample_rate            = 16000.0;
    c->win_len_s              = 0.025;
    c->win_shift_s            = 0.01;
    c->preemph_alpha          = 1.0;
    c->fft_max_input_s        = 0.025;
    c->apply_scale            = 1;
    c->mel_filter_width       = 268.258;
    c->mel_spacing            = 0.0;
    c->warp_differential_unit = 1;
    c->n_ceps                 = 16;
    c->dct_normalize          = 0;
}

int amx_mfcc_create(amx_ctx* ctx, const amx_mfcc_cfg* cfg, amx_mfcc** out) {
    // ctx == NULL creates a host-only handle: geometry and tables are available
    // (amx_mfcc_describe / _n_frames / _tables), running it returns AMX_ERR_STATE.
    AMX_REQUIRE(cfg && out, AMX_ERR_INVALID, "amx_mfcc_create: NULL argument");
    *out        = nullptr;
    amx_mfcc* h = new amx_mfcc;
    h->ctx      = ctx;
    int r       = h->tab.build(*cfg);
    if (r != AMX_OK) {
        delete h;
        return r;
    }
    const amx::MfccTables& t = h->tab;
    if (t.fft_len < 8 || t.fft_len > 4096) {
        amx::set_error("amx_mfcc_create: FFT length %d not supported by the gfx950 kernel (8..4096)", t.fft_len);
        delete h;
        return AMX_ERR_UNSUPPORTED;
    }
    if (!ctx) {
        *out = h;
        return AMX_OK;
    }
    AMX_HIP(hipSetDevice(ctx->device));
    // frames per tile: as many as keep the workgroup's LDS under ~40 KB (4 workgroups per CU)
    h->frames_per_tile = 32;
    while (h->frames_per_tile > 4 && mfcc_lds_bytes(t, h->frames_per_tile) > 40 * 1024)
        h->frames_per_tile /= 2;
    h->lds_bytes = mfcc_lds_bytes(t, h->frames_per_tile);
    if (h->lds_bytes > 160 * 1024) {
        amx::set_error("amx_mfcc_create: configuration needs %zu bytes of LDS per workgroup (> 160 KiB)", h->lds_bytes);
        delete h;
        return AMX_ERR_UNSUPPORTED;
    }
    std::vector<float> dct_t((size_t)t.n_filters * t.n_ceps);
    for (int k = 0; k < t.n_ceps; ++k)
        for (int n = 0; n < t.n_filters; ++n)
            dct_t[(size_t)n * t.n_ceps + k] = t.dct[(size_t)k * t.n_filters + n];
    if ((r = upload(&h->d_window, t.window.data(), t.window.size())) != AMX_OK ||
        (r = upload(&h->d_fw, t.filter_weights.data(), t.filter_weights.size())) != AMX_OK ||
        (r = upload(&h->d_dct_t, dct_t.data(), dct_t.size())) != AMX_OK ||
        (r = upload(&h->d_fs, t.filter_start.data(), t.filter_start.size())) != AMX_OK ||
        (r = upload(&h->d_fe, t.filter_end.data(), t.filter_end.size())) != AMX_OK ||
        (r = upload(&h->d_fo, t.filter_offset.data(), t.filter_offset.size())) != AMX_OK ||
        (r = upload(&h->d_tw, (const float2*)t.twiddle.data(), t.twiddle.size() / 2)) != AMX_OK ||
        (r = upload(&h->d_stw, (const float2*)t.split_twiddle.data(), t.split_twiddle.size() / 2)) != AMX_OK) {
        amx_mfcc_destroy(h);
        return r;
    }
    *out = h;
    return AMX_OK;
}

void amx_mfcc_destroy(amx_mfcc* h) {
    if (!h)
        return;
    if (!h->ctx) {
        delete h;
        return;
    }
    hipSetDevice(h->ctx->device);
    hipFree(h->d_window);
    hipFree(h->d_fw);
    hipFree(h->d_dct_t);
    hipFree(h->d_fs);
    hipFree(h->d_fe);
    hipFree(h->d_fo);
    hipFree(h->d_tw);
    hipFree(h->d_stw);
    delete h;
}

int amx_mfcc_describe(const amx_mfcc* h, amx_mfcc_info* info) {
    AMX_REQUIRE(h && info, AMX_ERR_INVALID, "amx_mfcc_describe: NULL argument");
    info->frame_len              = h->tab.frame_len;
    info->frame_shift            = h->tab.frame_shift;
    info->fft_len                = h->tab.fft_len;
    info->n_bins                 = h->tab.n_bins;
    info->n_filters              = h->tab.n_filters;
    info->n_ceps                 = h->tab.n_ceps;
    info->fft_output_sample_rate = h->tab.fft_output_sample_rate;
    info->mel_max                = h->tab.mel_max;
    return AMX_OK;
}

long amx_mfcc_n_frames(const amx_mfcc* h, long n_samples) {
    return h ? h->tab.n_frames(n_samples) : 0;
}

double amx_mfcc_frame_start_time(const amx_mfcc* h, long frame) {
    return h ? h->tab.frame_start_time(frame) : 0.0;
}

int amx_mfcc_tables(const amx_mfcc* h, float* window, int* fs, int* fe, int* fo, float* fw, float* dct) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_mfcc_tables: NULL handle");
    const amx::MfccTables& t = h->tab;
    if (window)
        memcpy(window, t.window.data(), t.window.size() * 4);
    if (fs)
        memcpy(fs, t.filter_start.data(), t.filter_start.size() * 4);
    if (fe)
        memcpy(fe, t.filter_end.data(), t.filter_end.size() * 4);
    if (fo)
        memcpy(fo, t.filter_offset.data(), t.filter_offset.size() * 4);
    if (fw)
        memcpy(fw, t.filter_weights.data(), t.filter_weights.size() * 4);
    if (dct)
        memcpy(dct, t.dct.data(), t.dct.size() * 4);
    return AMX_OK;
}

int amx_mfcc_plan_create(amx_mfcc* h, int n_seg, const long* sample_offsets, amx_mfcc_plan** out) {
    AMX_REQUIRE(h && out && n_seg >= 0 && (n_seg == 0 || sample_offsets), AMX_ERR_INVALID, "amx_mfcc_plan_create: bad argument");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_mfcc_plan_create: host-only handle (created without a context)");
    *out             = nullptr;
    amx_mfcc_plan* p = new amx_mfcc_plan;
    p->owner         = h;
    p->n_seg         = n_seg;
    p->sample_off.assign(sample_offsets, sample_offsets + n_seg + (n_seg ? 1 : 0));
    if (n_seg == 0)
        p->sample_off.assign(1, 0);
    p->frame_off.assign((size_t)n_seg + 1, 0);
    const int ft = h->frames_per_tile;
    for (int u = 0; u < n_seg; ++u) {
        long len = p->sample_off[u + 1] - p->sample_off[u];
        if (len < 0 || len > 0x7fffffffL) {
            amx::set_error("amx_mfcc_plan_create: segment %d has invalid length %ld", u, len);
            delete p;
            return AMX_ERR_INVALID;
        }
        long T              = h->tab.n_frames(len);
        p->frame_off[u + 1] = p->frame_off[u] + T;
        for (long f0 = 0; f0 < T; f0 += ft) {
            amx::MfccTile t;
            t.sample_base = p->sample_off[u];
            t.out_frame   = p->frame_off[u] + f0;
            t.n_samples   = (int)len;
            t.frame0      = (int)f0;
            t.n_frames    = (int)std::min<long>(ft, T - f0);
            t.pad_        = 0;
            p->tiles.push_back(t);
        }
    }
    hipSetDevice(h->ctx->device);
    int r = upload(&p->d_tiles, p->tiles.data(), p->tiles.size());
    if (r == AMX_OK) {
        std::vector<long long> fo(p->frame_off.begin(), p->frame_off.end());
        r = upload(&p->d_frame_off, fo.data(), fo.size());
    }
    if (r != AMX_OK) {
        amx_mfcc_plan_destroy(p);
        return r;
    }
    *out = p;
    return AMX_OK;
}

void amx_mfcc_plan_destroy(amx_mfcc_plan* p) {
    if (!p)
        return;
    hipFree(p->d_tiles);
    hipFree(p->d_frame_off);
    delete p;
}

long amx_mfcc_plan_total_frames(const amx_mfcc_plan* p) {
    return p ? p->frame_off.back() : 0;
}

int amx_mfcc_plan_frame_offsets(const amx_mfcc_plan* p, long* frame_offsets) {
    AMX_REQUIRE(p && frame_offsets, AMX_ERR_INVALID, "amx_mfcc_plan_frame_offsets: NULL argument");
    std::copy(p->frame_off.begin(), p->frame_off.end(), frame_offsets);
    return AMX_OK;
}

int amx_mfcc_run_plan_dev(amx_mfcc* h, const amx_mfcc_plan* p, const float* pcm_dev, float* ceps_dev) {
    AMX_REQUIRE(h && p, AMX_ERR_INVALID, "amx_mfcc_run_plan_dev: NULL handle");
    AMX_REQUIRE(p->owner == h, AMX_ERR_STATE, "amx_mfcc_run_plan_dev: plan belongs to another front-end handle");
    if (p->tiles.empty())
        return AMX_OK;
    AMX_REQUIRE(pcm_dev && ceps_dev, AMX_ERR_INVALID, "amx_mfcc_run_plan_dev: NULL buffer");
    AMX_HIP(hipSetDevice(h->ctx->device));
    const amx::MfccTables& t = h->tab;
    amx::MfccParams        k;
    k.pcm             = pcm_dev;
    k.ceps            = ceps_dev;
    k.tiles           = p->d_tiles;
    k.window          = h->d_window;
    k.fstart          = h->d_fs;
    k.fend            = h->d_fe;
    k.foff            = h->d_fo;
    k.fweights        = h->d_fw;
    k.dct_t           = h->d_dct_t;
    k.tw              = h->d_tw;
    k.stw             = h->d_stw;
    k.frame_len       = t.frame_len;
    k.frame_shift     = t.frame_shift;
    k.n_filters       = t.n_filters;
    k.n_ceps          = t.n_ceps;
    k.n_weights       = (int)t.filter_weights.size();
    k.frames_per_tile = h->frames_per_tile;
    k.alpha           = (float)t.cfg.preemph_alpha;
    k.fft_scale       = t.fft_scale;
    k.apply_scale     = (t.cfg.apply_scale && t.cfg.sample_rate != 1) ? 1 : 0;
    k.dct_normalize   = t.cfg.dct_normalize;
    const int n_tiles = (int)p->tiles.size();
    switch (t.fft_len / 2) {
        case 4: return launch_mfcc<4>(h, k, n_tiles);
        case 8: return launch_mfcc<8>(h, k, n_tiles);
        case 16: return launch_mfcc<16>(h, k, n_tiles);
        case 32: return launch_mfcc<32>(h, k, n_tiles);
        case 64: return launch_mfcc<64>(h, k, n_tiles);
        case 128: return launch_mfcc<128>(h, k, n_tiles);
        case 256: return launch_mfcc<256>(h, k, n_tiles);
        case 512: return launch_mfcc<512>(h, k, n_tiles);
        case 1024: return launch_mfcc<1024>(h, k, n_tiles);
        case 2048: return launch_mfcc<2048>(h, k, n_tiles);
        default:
            amx::set_error("amx_mfcc_run_plan_dev: no kernel for FFT length %d", t.fft_len);
            return AMX_ERR_UNSUPPORTED;
    }
}

int amx_mfcc_run_batch(amx_mfcc* h, int n_seg, const float* const* pcm_host, const long* n_samples, float* const* ceps_host) {
    AMX_REQUIRE(h && n_seg >= 0, AMX_ERR_INVALID, "amx_mfcc_run_batch: bad argument");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_mfcc_run_batch: host-only handle (created without a context)");
    if (n_seg == 0)
        return AMX_OK;
    AMX_REQUIRE(pcm_host && n_samples && ceps_host, AMX_ERR_INVALID, "amx_mfcc_run_batch: NULL argument");
    std::vector<long> off((size_t)n_seg + 1, 0);
    for (int u = 0; u < n_seg; ++u) {
        AMX_REQUIRE(n_samples[u] >= 0, AMX_ERR_INVALID, "amx_mfcc_run_batch: negative segment length");
        off[u + 1] = off[u] + n_samples[u];
    }
    amx_mfcc_plan* plan = nullptr;
    int            r    = amx_mfcc_plan_create(h, n_seg, off.data(), &plan);
    if (r != AMX_OK)
        return r;
    const long total_frames = amx_mfcc_plan_total_frames(plan);
    float *    d_pcm = nullptr, *d_ceps = nullptr;
    hipStream_t st = h->ctx->stream;
    auto fail = [&](int code) {
        hipFree(d_pcm);
        hipFree(d_ceps);
        amx_mfcc_plan_destroy(plan);
        return code;
    };
    if (hipMalloc((void**)&d_pcm, std::max<long>(off[n_seg], 1) * 4) != hipSuccess ||
        hipMalloc((void**)&d_ceps, std::max<long>(total_frames * h->tab.n_ceps, 1) * 4) != hipSuccess) {
        amx::set_error("amx_mfcc_run_batch: out of device memory");
        return fail(AMX_ERR_DEVICE);
    }
    for (int u = 0; u < n_seg; ++u)
        if (n_samples[u] > 0 &&
            hipMemcpyAsync(d_pcm + off[u], pcm_host[u], (size_t)n_samples[u] * 4, hipMemcpyHostToDevice, st) != hipSuccess) {
            amx::set_error("amx_mfcc_run_batch: H2D copy failed");
            return fail(AMX_ERR_DEVICE);
        }
    r = amx_mfcc_run_plan_dev(h, plan, d_pcm, d_ceps);
    if (r != AMX_OK)
        return fail(r);
    for (int u = 0; u < n_seg; ++u) {
        long T = plan->frame_off[u + 1] - plan->frame_off[u];
        if (T > 0 && hipMemcpyAsync(ceps_host[u], d_ceps + plan->frame_off[u] * h->tab.n_ceps, (size_t)T * h->tab.n_ceps * 4,
                                    hipMemcpyDeviceToHost, st) != hipSuccess) {
            amx::set_error("amx_mfcc_run_batch: D2H copy failed");
            return fail(AMX_ERR_DEVICE);
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess) {
        amx::set_error("amx_mfcc_run_batch: kernel execution failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(AMX_ERR_DEVICE);
    }
    return fail(AMX_OK);
}

int amx_mfcc_run(amx_mfcc* h, const float* pcm_host, long n_samples, float* ceps_host) {
    const float* in[1]  = {pcm_host};
    float*       out[1] = {ceps_host};
    long         n[1]   = {n_samples};
    return amx_mfcc_run_batch(h, 1, in, n, out);
}

int amx_context_window_dev(amx_ctx* ctx, const amx_mfcc_plan* p, const float* feats_dev, int dim, int left, int right,
                           float* out_dev, int out_stride) {
    AMX_REQUIRE(ctx && p && feats_dev && out_dev, AMX_ERR_INVALID, "amx_context_window_dev: NULL argument");
    AMX_REQUIRE(dim > 0 && left >= 0 && right >= 0 && out_stride >= (left + right + 1) * dim, AMX_ERR_INVALID,
                "amx_context_window_dev: out_stride %d < window %d", out_stride, (left + right + 1) * dim);
    const long long total = p->frame_off.back();
    if (total == 0)
        return AMX_OK;
    AMX_HIP(hipSetDevice(ctx->device));
    amx::ScopedKernelTimer timer(ctx, "context_window");
    hipLaunchKernelGGL(amx::context_window_kernel, dim3((unsigned)total), dim3(256), 0, ctx->stream, feats_dev,
                       p->d_frame_off, p->n_seg, dim, left, right, out_dev, out_stride, total);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

}  // extern "C"
