// gmm_tied.hip -- pruned exact maximum-approximation scorer for tied models whose mixtures all list the same densities
// (Mm::GaussDiagonalMaximumFeatureScorer on a tied-mixture set: Mm/GaussDiagonalMaximumFeatureScorer.cc:116-141).
//
// score(t, m) = min_k s_k with s_k = (m2lw[m][k] + logNorm[k]) + dist[k][t] in f64 and the sequential rule
// `if ((double)best > s) { best = (float)s; idx = k; }`.  gmm_tied_tile_kernel (gmm.hip) evaluates the (min,+) product
// densely: K x n_mix x T sums, twice.  But a density far from the frame cannot win in ANY mixture, and that can be proven
// per (density, frame, 64-mixture tile) from small tables:
//
//   a^[k][m]     = fl32(m2lw[k][m] + logNorm[k])             (formed on the fly from the weight table [K][mix_pad])
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
__global__ __launch_bounds__(256) void tied_bound_kernel(const float* __restrict__ g_m2lw_t, const float* __restrict__ g_ln32,
                                                        const float* __restrict__ g_amax, const float* __restrict__ g_dt, int K, int Kpad,
                                                        int n_mix, int mix_pad, int n_tiles, float* __restrict__ g_thr) {
    __shared__ float    s_v[256];
    __shared__ uint32_t s_i[256];
    __shared__ float    s_nd[kTiedNear], s_nl[kTiedNear];
    __shared__ uint32_t s_nk[kTiedNear];
    const int           t = blockIdx.y, tid = threadIdx.x, m = blockIdx.x * 256 + tid;
    const float*        row = g_dt + (size_t)t * Kpad;
    {
        float    bv = __builtin_inff();
        uint32_t bi = 0;
        for (int k = tid; k < K; k += 256) {  // 256 = 8 x 32: a thread stays inside one residue class
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
#pragma unroll
        for (int j = 1; j < 256 / kTiedNear; ++j)
            if (s_v[tid + kTiedNear * j] < bv) {
                bv = s_v[tid + kTiedNear * j];
                bi = s_i[tid + kTiedNear * j];
            }
        s_nd[tid] = bv;  // +inf: empty class, or no finite distance (NaN / inf frame)
        s_nk[tid] = bi;
        s_nl[tid] = g_ln32[bi];
    }
    __syncthreads();
    float u = FLT_MAX;
    if (m < mix_pad) {
#pragma unroll 8
        for (int i = 0; i < kTiedNear; ++i) {
            const float d = s_nd[i];
            if (d < __builtin_inff()) {  // wave-uniform
                const float a = g_m2lw_t[(size_t)s_nk[i] * mix_pad + m] + s_nl[i];
                u             = fminf(u, a + d);
            }
        }
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
                                                      float* __restrict__ g_ld, int* __restrict__ g_ln) {
    const int t = blockIdx.x, lane = threadIdx.x;
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
#pragma unroll 4
    for (int kb = 0; kb < K; kb += 64) {
        const int                k    = kb + lane;
        const float              dv   = row[k];
        const bool               rel  = k < K && (g_amin_all[k] + dv) <= thr;  // k < K: +inf <= thr when the threshold is +inf itself
        const unsigned long long mask = __ballot(rel);
        if (rel) {
            const int pos = n + __popcll(mask & ((1ull << lane) - 1ull));
            lk[pos]       = (uint32_t)k;
            ld[pos]       = dv;
        }
        n += __popcll(mask);
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
constexpr int kTiedSeg = 128;  // survivors buffered per wave; a fuller list is worked off and the scan resumes

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
    const int           tile = blockIdx.x, t = blockIdx.y * 4 + wave;
    if (t >= T)
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
        for (; ib < nl && n + 64 <= kTiedSeg; ib += 64) {
            const int      i  = ib + lane;
            const bool     in = i < nl;
            const uint32_t k  = in ? lk[i] : 0u;
            const float    dv = in ? ld[i] : 0.f;
            const bool     rel = in && (arow[k] + dv) <= Thr;
            const unsigned long long mask = __ballot(rel);
            if (rel) {
                const int pos   = n + __popcll(mask & ((1ull << lane) - 1ull));
                s_k[wave][pos] = k;
                s_d[wave][pos] = dv;
                s_l[wave][pos] = g_ln64[k];
            }
            n += __popcll(mask);
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
extern "C" int amx_internal_gmm_tied_create(int K, int n_mix, int mix_pad, const float* ahat_t_host, float** d_amin) {
    const int          n_tiles = mix_pad / 64, Kpad = (K + 63) & ~63;
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
                                           int mix_pad, const float* ln32, const float* amax, const float* m2lw_t, const double* ln64,
                                           const float* amin, void* workspace, float* scores, uint32_t* best,
                                           unsigned long long* survivors_dev) {
    if (T <= 0)
        return AMX_OK;
    const int    Kpad = (K + 63) & ~63, n_tiles = mix_pad / 64;
    const TiedWs w    = tied_ws(workspace, K, T, mix_pad);
    hipLaunchKernelGGL(amx::tied_transpose_kernel, dim3(Kpad / 64, (T + 63) / 64), dim3(256), 0, ctx->stream, dist_dev, k_dens_dev, K, Kpad, T,
                       Tpad, w.dt);
    hipLaunchKernelGGL(amx::tied_bound_kernel, dim3((mix_pad + 255) / 256, T), dim3(256), 0, ctx->stream, m2lw_t, ln32, amax, w.dt, K, Kpad,
                       n_mix, mix_pad, n_tiles, w.thr);
    hipLaunchKernelGGL(amx::tied_list_kernel, dim3(T), dim3(64), 0, ctx->stream, w.dt, amin + (size_t)n_tiles * Kpad, w.thr, K, Kpad, n_tiles,
                       w.lk, w.ld, w.ln);
    hipLaunchKernelGGL(amx::tied_pruned_kernel, dim3(n_tiles, (T + 3) / 4), dim3(256), 0, ctx->stream, w.lk, w.ld, w.ln, amin, w.thr, m2lw_t,
                       ln64, Kpad, T, n_mix, mix_pad, n_tiles, scores, best, survivors_dev);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}
