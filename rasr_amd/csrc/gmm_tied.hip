// gmm_tied.hip -- pruned exact maximum-approximation scorer for tied models whose mixtures all list the same densities
// (Mm::GaussDiagonalMaximumFeatureScorer on a tied-mixture set: Mm/GaussDiagonalMaximumFeatureScorer.cc:116-141).
//
// score(t, m) = min_k s_k with s_k = (m2lw[m][k] + logNorm[k]) + dist[k][t] in f64 and the sequential rule
// `if ((double)best > s) { best = (float)s; idx = k; }`.  gmm_tied_tile_kernel (gmm.hip) evaluates the (min,+) product
// densely: K x n_mix x T sums, twice.  But a density far from the frame cannot win in ANY mixture, and that can be proven
// per (density, frame, 64-mixture tile) from small tables:
//
//   a^[k][m]     = fl32(m2lw[k][m] + logNorm[k])             (model; the bound kernel reads a bf16 image rounded up, [K][mix_pad])
//   amin[j][k]   = min over the mixtures of tile j of a^[k][m] (model, [n_tiles][Kpad]);  aminG[k] = min over all tiles
//   U[t][m]      = min over 32 densities NEAR frame t (the closest density of each residue class k mod 32) of
//                  s^_k = fl32(a^[k][m] + dist[k][t])         -- an upper bound of min_k s^_k, because it is a minimum over a subset
//
// Candidates -- the densities that can influence (best, idx), i.e. those whose f64 sum rounds to the winning f32 value, see
// the subsequence argument in front of gmm_tied_tile_kernel -- satisfy s^_k <= min s^ + tau <= U[t][m] + tau', with
// tau' = 2^-21 (2 max_k|a^[.][m]| + |U|) (the bound of that comment, written for an upper bound of the minimum: |min s^| <=
// max(|U|, max|a^|) since dist >= 0).  With Thr[t][j] = max over the tile's mixtures of U + tau', and fl32 monotone,
//   fl32(amin[j][k] + dist[k][t]) > Thr[t][j]   ==>   no mixture of tile j has density k as a candidate for frame t
// and the same with aminG and ThrG[t] = max_j Thr[t][j] for the whole model.  Four small kernels:
//   tied_transpose_kernel   dist[k][t] -> dt[t][k] (list order), so that a frame's distances are one contiguous row
//   tied_bound_kernel       near densities of the frame, U, Thr[t][j]
//   tied_list_kernel        per frame: the densities that pass the model-wide test (4 % on the config-3 instance), ascending
//   tied_pruned_kernel      per (tile, frame): the tile test over that list (1.2 % of all densities pass), then the
//                           reference's own f64 rule over the survivors in ascending k for the tile's 64 mixtures (lane = mixture).
// Running the rule over a subsequence that contains every candidate, in the original order, is bit-identical.
//
// A model / feature distribution without that structure (everything survives) would make this slower than the dense
// kernel -- each table element is then used once instead of 16 times from registers.  The kernel counts the survivors; the host
// reads the count of earlier calls and goes back to gmm_tied_tile_kernel while the surviving fraction is high (gmm.hip).
#include "common.hpp"

#include <cfloat>
#include <cstring>
#include <vector>

namespace amx {

constexpr int kTiedNear     = 32;   // near densities per frame = residue classes of the density index
constexpr int kTiedCounters = 256;  // survivor counters (summed by the host)

// dist [n_dens][Tpad] (coalesced along frames) -> dt [T][Kpad] (coalesced along the density list); 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void tied_transpose_kernel(const float* __restrict__ g_dist, const uint32_t* __restrict__ g_k_dens, int K,
                                                           int Kpad, int T, int Tpad, float* __restrict__ g_dt) {
    __shared__ float s[64][65];
    const int        k0 = blockIdx.x * 64, t0 = blockIdx.y * 64;
    const int        c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = k0 + r0 + 4 * i;
        s[r0 + 4 * i][c] = (k < K && t0 + c < Tpad) ? g_dist[(size_t)g_k_dens[k] * Tpad + t0 + c] : __builtin_inff();
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int t = t0 + r0 + 4 * i;
        if (t < T)
            g_dt[(size_t)t * Kpad + k0 + c] = s[c][r0 + 4 * i];  // k >= K: +inf
    }
}

// Thr[t][tile]: thread = mixture, 256 mixtures = 4 tiles per workgroup.  Prologue: the frame's closest density of every residue
// class k mod 32 (any subset gives a valid bound; this one needs no selection).  U = min over them of fl32(a^ + dist); the tile's
// threshold is the maximum of U + tau' over its real mixtures.
// The sums only have to bound the minimum from ABOVE, so the 32 table rows per frame are read from a bf16 image of a^ that was
// rounded UP (a^_up >= a^, hence fl32(a^_up + dist) >= fl32(a^ + dist)): half the bytes of the kernel's only real traffic, for a
// bound that is at most 2^-8 |a^| looser.
constexpr int kTiedBoundThreads = 1024;  // mixtures per workgroup of tied_bound_kernel (the prologue is paid once per workgroup)

__global__ __launch_bounds__(kTiedBoundThreads) void tied_bound_kernel(const unsigned short* __restrict__ g_aup, const float* __restrict__ g_amax,
                                                                      const float* __restrict__ g_dt, int K, int Kpad, int n_mix, int mix_pad,
                                                                      int n_tiles, float* __restrict__ g_thr) {
    constexpr int       NT = kTiedBoundThreads;
    __shared__ float    s_v[NT];
    __shared__ uint32_t s_i[NT];
    __shared__ float    s_nd[kTiedNear];
    __shared__ uint32_t s_nk[kTiedNear];
    const int           t = blockIdx.y, tid = threadIdx.x, m = blockIdx.x * NT + tid;
    const float*        row = g_dt + (size_t)t * Kpad;
    {
        float    bv = __builtin_inff();
        uint32_t bi = 0;
        for (int k = tid; k < K; k += NT) {  // NT is a multiple of 32: a thread stays inside one residue class
            const float v = row[k];
            if (v < bv) {
                bv = v;
                bi = (uint32_t)k;
            }
        }
        s_v[tid] = bv;
        s_i[tid] = bi;
    }
    __syncthreads();
    if (tid < kTiedNear) {
        float    bv = s_v[tid];
        uint32_t bi = s_i[tid];
        for (int j = 1; j < NT / kTiedNear; ++j)
            if (s_v[tid + kTiedNear * j] < bv) {
                bv = s_v[tid + kTiedNear * j];
                bi = s_i[tid + kTiedNear * j];
            }
        s_nd[tid] = bv;  // +inf: empty class, or no finite distance (NaN / inf frame); row 0 stands in, its sum is +inf
        s_nk[tid] = bi;
    }
    __syncthreads();
    float u = FLT_MAX;
    if (m < mix_pad) {
        float a[kTiedNear];
#pragma unroll
        for (int i = 0; i < kTiedNear; ++i)  // all rows in flight before the first use
            a[i] = __uint_as_float((uint32_t)g_aup[(size_t)s_nk[i] * mix_pad + m] << 16);
#pragma unroll
        for (int i = 0; i < kTiedNear; ++i)
            u = fminf(u, a[i] + s_nd[i]);
    }
    float thr = -__builtin_inff();
    if (m < n_mix) {
        thr = u + (4.76837158e-7f * (2.f * g_amax[m] + fabsf(u)) + 1e-30f);  // tau' = 2^-21 (2 max|a^| + |U|)
        if (!(thr == thr))
            thr = __builtin_inff();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        thr = fmaxf(thr, __shfl_xor(thr, o));
    const int tile = m >> 6;
    if ((tid & 63) == 0 && tile < n_tiles)
        g_thr[(size_t)t * n_tiles + tile] = thr;
}

// One wave per frame: ThrG = max over the tiles, then the densities with fl32(aminG[k] + dist) <= ThrG, ascending, with their
// distances.  lk / ld [T][Kpad], ln [T].
__global__ __launch_bounds__(64) void tied_list_kernel(const float* __restrict__ g_dt, const float* __restrict__ g_amin_all,
                                                      const float* __restrict__ g_thr, int K, int Kpad, int n_tiles, uint32_t* __restrict__ g_lk,
                                                      float* __restrict__ g_ld, int* __restrict__ g_ln,
                                                      unsigned long long* __restrict__ g_examined, unsigned long long examined) {
    const int t = blockIdx.x, lane = threadIdx.x;
    // the denominator of the survivor statistic travels with the numerator (the host reads both from ONE asynchronous copy: counting
    // the submitted triples on the host instead made the ratio wrong whenever the host ran ahead of the device)
    if (g_examined && t == 0 && lane == 0)
        atomicAdd(g_examined, examined);
    float     thr = -__builtin_inff();
    for (int j = lane; j < n_tiles; j += 64)
        thr = fmaxf(thr, g_thr[(size_t)t * n_tiles + j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        thr = fmaxf(thr, __shfl_xor(thr, o));
    const float* row = g_dt + (size_t)t * Kpad;
    uint32_t*    lk  = g_lk + (size_t)t * Kpad;
    float*       ld  = g_ld + (size_t)t * Kpad;
    int          n   = 0;
    for (int kb = 0; kb < Kpad; kb += 256) {  // four 64-density steps per trip, their eight loads in flight together
        float dv[4], am[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kb + 64 * u + lane;
            dv[u]       = k < Kpad ? row[k] : __builtin_inff();
            am[u]       = k < Kpad ? g_amin_all[k] : __builtin_inff();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int                k    = kb + 64 * u + lane;
            const bool               rel  = k < K && (am[u] + dv[u]) <= thr;  // k < K: +inf <= thr when the threshold is +inf itself
            const unsigned long long mask = __ballot(rel);
            if (rel) {
                const int pos = n + __popcll(mask & ((1ull << lane) - 1ull));
                lk[pos]       = (uint32_t)k;
                ld[pos]       = dv[u];
            }
            n += __popcll(mask);
        }
    }
    if (lane == 0)
        g_ln[t] = n;
}

struct TiedMax {  // MaxState of gmm.hip (the reference's rule), restated here to keep this file self-contained
    float    best   = FLT_MAX;
    double   best_d = (double)FLT_MAX;
    uint32_t idx    = 0xffffffffu;
    __device__ __forceinline__ void add(double c64, float dist, uint32_t k) {
        const double s = c64 + (double)dist;
        if (best_d > s) {
            best   = (float)s;
            best_d = (double)best;
            idx    = k;
        }
    }
};

// One wave per (64-mixture tile, frame).  Phase 1 (lane = list entry) applies the tile's test to the frame's list and compacts
// the survivors -- position, distance, log-normalisation term -- into LDS; phase 2 (lane = mixture) runs the f64 rule over them
// in ascending order with PF rows of the weight table in flight.
constexpr int kTiedSeg = 256;  // survivors buffered per wave; a fuller list is worked off and the scan resumes

__global__ __launch_bounds__(256) void tied_pruned_kernel(const uint32_t* __restrict__ g_lk, const float* __restrict__ g_ld,
                                                         const int* __restrict__ g_ln, const float* __restrict__ g_amin,
                                                         const float* __restrict__ g_thr, const float* __restrict__ g_m2lw_t,
                                                         const double* __restrict__ g_ln64, int Kpad, int T, int n_mix, int mix_pad,
                                                         int n_tiles, float* __restrict__ g_scores, uint32_t* __restrict__ g_best,
                                                         unsigned long long* __restrict__ g_survivors) {
    __shared__ uint32_t s_k[4][kTiedSeg];
    __shared__ float    s_d[4][kTiedSeg];
    __shared__ double   s_l[4][kTiedSeg];
    const int           lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Workgroup b runs on XCD b % 8 (each XCD has its own L2).  A row of the weight table is wanted by ~3 of a 256-frame batch's
    // frames, so all frame groups of one tile go to ONE XCD, back to back: tile = 8 * (slot / n_fg) + xcd, frame group = slot % n_fg.
    // The tile's 1 MB slice of the table then comes from HBM once instead of once per frame that wants it.
    const int n_fg = (T + 3) / 4, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = (slot / n_fg) * 8 + xcd, t = (slot % n_fg) * 4 + wave;
    if (tile >= n_tiles || t >= T)
        return;
    const int       m    = tile * 64 + lane;
    const float     Thr  = g_thr[(size_t)t * n_tiles + tile];
    const float*    arow = g_amin + (size_t)tile * Kpad;
    const uint32_t* lk   = g_lk + (size_t)t * Kpad;
    const float*    ld   = g_ld + (size_t)t * Kpad;
    const int       nl   = g_ln[t];
    TiedMax         st;
    int             total = 0;
    int             ib    = 0;
    while (ib < nl) {
        // ---- phase 1
        int n = 0;
        for (; ib < nl && n + 128 <= kTiedSeg; ib += 128) {  // two 64-entry steps per trip: list loads, then gathers, then the tests
            uint32_t k[2];
            float    dv[2], am[2];
            double   ln[2];
            bool     in[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = ib + 64 * u + lane;
                in[u]       = i < nl;
                k[u]        = in[u] ? lk[i] : 0u;
                dv[u]       = in[u] ? ld[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                am[u] = arow[k[u]];
                ln[u] = g_ln64[k[u]];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool               rel  = in[u] && (am[u] + dv[u]) <= Thr;
                const unsigned long long mask = __ballot(rel);
                if (rel) {
                    const int pos  = n + __popcll(mask & ((1ull << lane) - 1ull));
                    s_k[wave][pos] = k[u];
                    s_d[wave][pos] = dv[u];
                    s_l[wave][pos] = ln[u];
                }
                n += __popcll(mask);
            }
        }
        total += n;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 2
        constexpr int PF = 16;
        for (int i = 0; i < n; i += PF) {
            float w[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int ii = i + j < n ? i + j : n - 1;
                w[j]         = g_m2lw_t[(size_t)s_k[wave][ii] * mix_pad + m];
            }
#pragma unroll
            for (int j = 0; j < PF; ++j)
                if (i + j < n)
                    st.add((double)w[j] + s_l[wave][i + j], s_d[wave][i + j], s_k[wave][i + j]);
        }
        __builtin_amdgcn_wave_barrier();  // the lists are rewritten by the next segment
    }
    if (m < n_mix) {
        g_scores[(size_t)t * n_mix + m] = 0.5f * st.best;
        if (g_best)
            g_best[(size_t)t * n_mix + m] = st.idx;
    }
    // statistics for the host's dense / pruned decision: spread over kTiedCounters addresses (40 000 atomics on ONE address cost
    // 0.37 ms, more than the rest of this kernel)
    if (lane == 0 && g_survivors)
        atomicAdd(g_survivors + ((tile * 7 + t) & (kTiedCounters - 1)), (unsigned long long)total);
}

}  // namespace amx

// amin[tile][k] = min over the real mixtures of the tile of a^[k][m]; aminG[k] = min over the tiles.  Returns one device table
// [(n_tiles + 1)][Kpad] (+inf padded), row n_tiles = aminG.
// bf16 that is >= the f32 value (NaN / inf pass through; +0 for the padding columns)
static unsigned short tied_bf16_up(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7f800000u) == 0x7f800000u || (u & 0xffffu) == 0)
        return (unsigned short)(u >> 16);
    // positive: truncation rounds down -> next bf16 up; negative: truncation (towards zero) already rounds up
    return (unsigned short)((u >> 16) + ((u >> 31) ? 0u : 1u));
}

extern "C" int amx_internal_gmm_tied_create(int K, int n_mix, int mix_pad, const float* ahat_t_host, float** d_amin, unsigned short** d_aup) {
    const int          n_tiles = mix_pad / 64, Kpad = (K + 63) & ~63;
    {
        std::vector<unsigned short> up((size_t)K * mix_pad);
        for (size_t i = 0; i < up.size(); ++i)
            up[i] = tied_bf16_up(ahat_t_host[i]);
        *d_aup = nullptr;
        AMX_HIP(hipMalloc((void**)d_aup, up.size() * 2));
        AMX_HIP(hipMemcpy(*d_aup, up.data(), up.size() * 2, hipMemcpyHostToDevice));
    }
    std::vector<float> amin((size_t)(n_tiles + 1) * Kpad, __builtin_inff());
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < n_mix; ++i) {
            const float a = ahat_t_host[(size_t)k * mix_pad + i];
            float&      v = amin[(size_t)(i >> 6) * Kpad + k];
            v             = a < v ? a : v;
            float& g      = amin[(size_t)n_tiles * Kpad + k];
            g             = a < g ? a : g;
        }
    *d_amin = nullptr;
    AMX_HIP(hipMalloc((void**)d_amin, amin.size() * sizeof(float)));
    AMX_HIP(hipMemcpy(*d_amin, amin.data(), amin.size() * sizeof(float), hipMemcpyHostToDevice));
    return AMX_OK;
}

namespace {
struct TiedWs {
    float *   dt, *ld, *thr;
    uint32_t* lk;
    int*      ln;
    size_t    bytes;
};
TiedWs tied_ws(void* base, int K, int T, int mix_pad) {
    const size_t Kpad = (size_t)((K + 63) & ~63), n_tiles = (size_t)mix_pad / 64;
    auto         al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    char*        p  = (char*)base;
    TiedWs       w;
    w.dt = (float*)p;
    p += al((size_t)T * Kpad * 4);
    w.lk = (uint32_t*)p;
    p += al((size_t)T * Kpad * 4);
    w.ld = (float*)p;
    p += al((size_t)T * Kpad * 4);
    w.ln = (int*)p;
    p += al((size_t)T * 4);
    w.thr = (float*)p;
    p += al((size_t)T * n_tiles * 4);
    w.bytes = (size_t)(p - (char*)base);
    return w;
}
}  // namespace

extern "C" size_t amx_internal_gmm_tied_workspace(int K, int T, int mix_pad) {
    return tied_ws(nullptr, K, T, mix_pad).bytes;
}

extern "C" int amx_internal_gmm_tied_score(amx_ctx* ctx, const float* dist_dev, const uint32_t* k_dens_dev, int K, int T, int Tpad, int n_mix,
                                           int mix_pad, const unsigned short* aup, const float* amax, const float* m2lw_t, const double* ln64,
                                           const float* amin, void* workspace, float* scores, uint32_t* best,
                                           unsigned long long* survivors_dev) {
    if (T <= 0)
        return AMX_OK;
    const int    Kpad = (K + 63) & ~63, n_tiles = mix_pad / 64;
    const TiedWs w    = tied_ws(workspace, K, T, mix_pad);
    hipLaunchKernelGGL(amx::tied_transpose_kernel, dim3(Kpad / 64, (T + 63) / 64), dim3(256), 0, ctx->stream, dist_dev, k_dens_dev, K, Kpad, T,
                       Tpad, w.dt);
    hipLaunchKernelGGL(amx::tied_bound_kernel, dim3((mix_pad + amx::kTiedBoundThreads - 1) / amx::kTiedBoundThreads, T), dim3(amx::kTiedBoundThreads), 0, ctx->stream, aup, amax, w.dt, K, Kpad, n_mix,
                       mix_pad, n_tiles, w.thr);
    hipLaunchKernelGGL(amx::tied_list_kernel, dim3(T), dim3(64), 0, ctx->stream, w.dt, amin + (size_t)n_tiles * Kpad, w.thr, K, Kpad, n_tiles,
                       w.lk, w.ld, w.ln, survivors_dev ? survivors_dev + amx::kTiedCounters : nullptr,
                       (unsigned long long)K * (unsigned long long)T * (unsigned long long)n_tiles);
    hipLaunchKernelGGL(amx::tied_pruned_kernel, dim3(8 * ((n_tiles + 7) / 8) * ((T + 3) / 4)), dim3(256), 0, ctx->stream, w.lk, w.ld, w.ln, amin, w.thr, m2lw_t,
                       ln64, Kpad, T, n_mix, mix_pad, n_tiles, scores, best, survivors_dev);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}
