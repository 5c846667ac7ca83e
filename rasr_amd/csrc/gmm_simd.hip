// gmm_simd.hip -- Mm::SimdGaussDiagonalMaximumFeatureScorer ("SIMD-diagonal-maximum") on gfx950.
//
// Reference: Mm/SimdFeatureScorer.cc:68-176, Mm/IntelOptimization.cc:37-66, quantize<f32,u8> (Mm/Utilities.hh:190-202),
// CovarianceFeatureScorerElement::scale (Mm/CovarianceFeatureScorerElement.cc:46-52).  Means and features are multiplied by
// scaling / sigma and quantised to u8 (round half away from zero, +128, clipped), the per-density constant is truncated to
// s32, and a density's score is constant + sum_i (mean_i - feature_i)^2 in integer arithmetic; the mixture takes the FIRST
// minimum and returns (f32)(0.5 * min / scaling^2) computed in f64.  Everything after the quantiser is exact integer work,
// so any evaluation order gives the reference's bits.
//
// Two paths:
//  * pooled covariance, <= 16 densities per mixture, dim <= 64 (the class's own prerequisite "#densities >> #covariances"):
//    with a' = mean - 128 and b' = feature - 128 (both fit i8) the distance is |a'|^2 + |b'|^2 - 2 a'.b', and a'.b' for 256
//    slots x 256 frames is an i8 MFMA product (v_mfma_i32_32x32x32_i8, K = 64 = one 64-byte row per slot / frame).  The
//    epilogue folds constant + |a'|^2 and the slot number into one key, key = ((c + |a'|^2) << 4 | slot) - 32 a'.b', so the
//    first minimum of a mixture is one v_mad_i32_i24 per density plus a v_min3 tree; |b'|^2 is added after the minimum.
//  * everything else (per-density covariances -- one quantised feature vector per covariance --, tied mixtures, long
//    mixtures): integer distances dist[density][frame] once per density, then a (min,+) pass over the mixture lists.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace amx {

typedef int simd_i32x4 __attribute__((ext_vector_type(4)));
typedef int simd_i32x16 __attribute__((ext_vector_type(16)));

// quantize<f32, u8>: clip((int)round(v) + 128).  (int) of a value outside the int range is undefined in C; the reference's
// x86-64 build gets INT_MIN from cvttsd2si (NaN included) and the +128 wraps, which is what this reproduces.
__device__ __forceinline__ int simd_quantize(float v) {
    const float r  = roundf(v);
    const int   qi = (r >= -2147483648.f && r < 2147483648.f) ? (int)r : INT_MIN;
    const int   q  = (int)((unsigned)qi + 128u);
    return min(max(q, 0), 255);
}

// (f32)(0.5 * minScore / scalingSquared) with the division in f64 like the reference.  0.5 * m / s2 = m / b with b = 2 * s2 (exact),
// and the correctly rounded quotient comes from the host's y = RN(1 / b) in three operations: q = RN(m y) is a faithful
// quotient, r = m - q b is exact in an fma, RN(q + r y) is RN(m / b) (Markstein; the excluded case, a divisor whose
// significand is all ones, cannot occur: b is an f32 value widened to f64).  A billion random (m, scaling) pairs agree with
// the IEEE division; the device's own v_div sequence is four times as long.
struct SimdScale {
    double b, y;
    float  int_scale;  // > 0: Mm::BatchIntFeatureScorer's conversion, (f32)min / scale_ in f32 (Mm/BatchFeatureScorer.cc:500)
};
__device__ __forceinline__ float simd_score(int min_score, SimdScale sc) {
    if (sc.int_scale > 0.f)
        return (float)min_score / sc.int_scale;
    const double a = (double)min_score;
    const double q = a * sc.y;
    const double r = fma(-q, sc.b, a);
    return (float)fma(r, sc.y, q);
}

// ---- pooled path, stage 1: one wavefront per frame, lane = dimension: X[t][64] i8 (feature - 128, zero padded), nx[t]
__global__ __launch_bounds__(256) void simd_quantize_kernel(const float* __restrict__ feats, const float* __restrict__ isr,
                                                           int8_t* __restrict__ X, int* __restrict__ nx, int T, int Tpad, int dim) {
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= Tpad)
        return;
    int b = 0;
    if (t < T && lane < dim)
        b = simd_quantize(feats[(size_t)t * dim + lane] * isr[lane]) - 128;
    X[(size_t)t * 64 + lane] = (int8_t)b;
    int s = b * b;
    for (int o = 32; o > 0; o >>= 1)
        s += __shfl_xor(s, o);
    if (lane == 0)
        nx[t] = s;
}

// ---- pooled path, stage 2: 256 slots x 256 frames per tile, 8 waves of 32 frames each (a wave sees all 16 mixtures of a slot
// tile for its frames, so a frame's 16 scores leave as one 64-byte piece), the frame tile resident in LDS -- its fragments in
// registers --, slot tiles and their keys double-buffered by LDS-DMA.  Rows are 64 bytes = four 16-byte chunks, chunk c of row
// r stored at position c ^ ((r >> 1) & 3): the eight lanes a ds_read_b128 serves per cycle then cover all 128 bytes of banks.
// A slot tile lists 16 mixtures x 16 slots with slot s of mixture 2b + h in row b*32 + (s>>2)*8 + h*4 + (s&3): in the
// 32x32 accumulator layout (row = (q>>2)*8 + (lane>>5)*4 + (q&3), column = lane & 31) lane half h then holds exactly the 16
// slots of mixture 2b + h for its frame, in slot order.
// Keys are stored negated, nkey = -(((c + |a'|^2) << 4) | slot): v_lshl_add_u32(dot, 5, nkey) = -(key - 32 dot) is one
// operation per density and the first minimum becomes a v_max3 tree (a smaller slot number gives the larger negated key).
constexpr int kSimdTile = 16 * 1024;
constexpr int kSimdLds  = 3 * kSimdTile + 2 * 1024 + 8 * (32 * 17 * 4 + 32 * 16);

__global__ __launch_bounds__(512, 2) void simd_mfma_kernel(const int8_t* __restrict__ g_A, const int8_t* __restrict__ g_X,
                                                          const int* __restrict__ g_nkey, const int* __restrict__ g_nx,
                                                          float* __restrict__ g_scores, uint32_t* __restrict__ g_best, int T, int n_mix,
                                                          int n_tiles_r, int r_split, SimdScale scale) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char*     s_x   = lds;
    char*     s_a   = lds + kSimdTile;                    // two buffers
    int*      s_key = (int*)(lds + 3 * kSimdTile);        // two buffers of 256 negated keys, [mixture][slot] order
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float*         s_sc = (float*)(lds + 3 * kSimdTile + 2048) + wave * 32 * 17;                              // [32 frames][17]
    unsigned char* s_bd = (unsigned char*)(lds + 3 * kSimdTile + 2048 + 8 * 32 * 17 * 4) + wave * 32 * 16;   // [32 frames][16]
    const int tile_t = blockIdx.x / r_split, part = blockIdx.x % r_split;
    const int per = (n_tiles_r + r_split - 1) / r_split;
    const int r_begin = part * per, r_end = min(n_tiles_r, r_begin + per);
    if (r_begin >= r_end)
        return;
    const int t0 = tile_t * 256;
    // LDS-DMA: lane l of wave w fills LDS bytes [pass*8192 + w*1024 + l*16, +16) = row (pass*128 + w*16 + (l>>2)), position
    // l & 3; the global chunk that belongs there is (l & 3) ^ ((row >> 1) & 3)
    auto load_tile = [&](const int8_t* base, char* dst) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 128 + wave * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((row >> 1) & 3);
            __builtin_amdgcn_global_load_lds((const void*)(base + (size_t)row * 64 + chunk * 16),
                                             (__attribute__((address_space(3))) void*)(dst + pass * 8192 + wave * 1024), 16, 0, 0);
        }
    };
    auto load_keys = [&](int r, int buf) {
        if (wave < 4)
            __builtin_amdgcn_global_load_lds((const void*)(g_nkey + (size_t)r * 256 + wave * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(s_key + buf * 256 + wave * 64), 4, 0, 0);
    };
    load_tile(g_X + (size_t)t0 * 64, s_x);
    load_tile(g_A + (size_t)r_begin * 256 * 64, s_a);
    load_keys(r_begin, 0);
    const int frow = lane & 31, fk = lane >> 5;
    const int nxv  = g_nx[t0 + wave * 32 + frow];
    simd_i32x4 bx[2];  // this wave's frame fragments: the frame tile does not change, so they stay in registers
    for (int r = r_begin; r < r_end; ++r) {
        const int buf = (r - r_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // my share of tile r (and its keys) is in LDS
        __builtin_amdgcn_s_barrier();                                // ... everybody's is, and buffer buf^1 is free again
        if (r + 1 < r_end) {
            load_tile(g_A + (size_t)(r + 1) * 256 * 64, s_a + (buf ^ 1) * kSimdTile);
            load_keys(r + 1, buf ^ 1);
        }
        if (r == r_begin) {
            const int rr = wave * 32 + frow;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                bx[ks] = *(const simd_i32x4*)(s_x + rr * 64 + (((ks * 2 + fk) ^ ((rr >> 1) & 3)) << 4));
        }
        const char* abase = s_a + buf * kSimdTile;
        const int*  keys  = s_key + buf * 256;
        // one 32-slot block (two mixtures) at a time: its two MFMAs, then its epilogue, so that the matrix pipe works on block
        // i + 1 while the vector pipe reduces block i.  Lane half fk owns mixture i*2 + fk of this tile for frame wave*32 + frow.
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int        rr = i * 32 + frow;
            const simd_i32x4 a0 = *(const simd_i32x4*)(abase + rr * 64 + ((fk ^ ((rr >> 1) & 3)) << 4));
            const simd_i32x4 a1 = *(const simd_i32x4*)(abase + rr * 64 + (((2 + fk) ^ ((rr >> 1) & 3)) << 4));
            simd_i32x16      acc;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                acc[q] = 0;
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bx[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bx[1], acc, 0, 0, 0);
            const int        ml = i * 2 + fk;
            const simd_i32x4 k0 = *(const simd_i32x4*)(keys + ml * 16), k1 = *(const simd_i32x4*)(keys + ml * 16 + 4);
            const simd_i32x4 k2 = *(const simd_i32x4*)(keys + ml * 16 + 8), k3 = *(const simd_i32x4*)(keys + ml * 16 + 12);
            const int        kq[16] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w, k2.x, k2.y, k2.z, k2.w, k3.x, k3.y, k3.z, k3.w};
            int              mx = INT_MIN;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                mx = max(mx, (int)(((unsigned)acc[q] << 5) + (unsigned)kq[q]));
            float    sc;
            unsigned bd;
            if (mx == INT_MIN) {  // a mixture without densities: the reference's initial values
                sc = simd_score(INT_MAX, scale);
                bd = 0xffu;
            }
            else {
                const int key = -mx;
                sc            = simd_score((key >> 4) + nxv, scale);
                bd            = (unsigned)key & 15u;
            }
            s_sc[frow * 17 + ml] = sc;
            s_bd[frow * 16 + ml] = (unsigned char)bd;
        }
        __builtin_amdgcn_wave_barrier();
        // two lanes per frame, 8 consecutive mixtures each: a frame's 64 bytes of scores leave in one instruction
        const int fl = lane >> 1, hf = lane & 1;
        const int t = t0 + wave * 32 + fl, m_first = r * 16 + hf * 8;
        if (t < T && m_first < n_mix) {
            const int nm = min(8, n_mix - m_first);
            float*    dst = g_scores + (size_t)t * n_mix + m_first;
            float     v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[q] = s_sc[fl * 17 + hf * 8 + q];
            if (nm == 8 && (((size_t)dst & 15) == 0)) {
                *(float4*)dst       = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            else
                for (int q = 0; q < nm; ++q)
                    dst[q] = v[q];
            if (g_best) {
                uint32_t*   bdst = g_best + (size_t)t * n_mix + m_first;
                const uint2 pk   = *(const uint2*)(s_bd + fl * 16 + hf * 8);
                uint32_t    w[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const unsigned c = ((q < 4 ? pk.x : pk.y) >> ((q & 3) * 8)) & 0xffu;
                    w[q]             = c == 0xffu ? 0xffffffffu : c;
                }
                if (nm == 8 && (((size_t)bdst & 15) == 0)) {
                    *(uint4*)bdst       = make_uint4(w[0], w[1], w[2], w[3]);
                    *(uint4*)(bdst + 4) = make_uint4(w[4], w[5], w[6], w[7]);
                }
                else
                    for (int q = 0; q < nm; ++q)
                        bdst[q] = w[q];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- general path, stage 1: dist[d][t] = sum_i (qmean[d][i] - quantize(x[t][i] * isr[cov(d)][i]))^2, lane = frame
__global__ __launch_bounds__(256) void simd_dist_kernel(const float* __restrict__ feats, const float* __restrict__ isr,
                                                       const unsigned char* __restrict__ qmean, const uint32_t* __restrict__ d_cov,
                                                       int* __restrict__ dist, int T, int ld, int dim, int n_dens, int dens_tile) {
    const int  t    = blockIdx.y * 256 + threadIdx.x;
    const bool live = t < T;
    const float* x  = feats + (size_t)(live ? t : T - 1) * dim;
    const int d0 = blockIdx.x * dens_tile, d1 = min(d0 + dens_tile, n_dens);
    for (int d = d0; d < d1; ++d) {
        const float*         is = isr + (size_t)d_cov[d] * dim;
        const unsigned char* qm = qmean + (size_t)d * dim;
        int                  s  = 0;
        for (int i = 0; i < dim; ++i) {
            const int df = (int)qm[i] - simd_quantize(x[i] * is[i]);
            s += df * df;
        }
        if (live)
            dist[(size_t)d * ld + t] = s;
    }
}

// ---- general path, stage 2: first minimum of cst[k] + dist[dens(k)][t] over each mixture's list
__global__ __launch_bounds__(256) void simd_combine_kernel(const int* __restrict__ dist, const int* __restrict__ cst,
                                                          const uint32_t* __restrict__ mix_off, const uint32_t* __restrict__ k_dens,
                                                          float* __restrict__ scores, uint32_t* __restrict__ best, int T, int ld, int n_mix,
                                                          int mix_tile, SimdScale scale) {
    const int  t    = blockIdx.y * 256 + threadIdx.x;
    const bool live = t < T;
    const int  tt   = live ? t : T - 1;
    const int m0 = blockIdx.x * mix_tile, m1 = min(m0 + mix_tile, n_mix);
    for (int m = m0; m < m1; ++m) {
        const uint32_t k0 = mix_off[m], k1 = mix_off[m + 1];
        int            mn = INT_MAX;
        uint32_t       bi = 0xffffffffu;
        for (uint32_t k = k0; k < k1; ++k) {
            const int s = cst[k] + dist[(size_t)k_dens[k] * ld + tt];
            if (s < mn) {
                mn = s;
                bi = k - k0;
            }
        }
        if (live) {
            scores[(size_t)t * n_mix + m] = simd_score(mn, scale);
            if (best)
                best[(size_t)t * n_mix + m] = bi;
        }
    }
}

// ---- Mm::BatchPreselectionIntFeatureScorer ("preselection-batch-int", Mm/BatchFeatureScorer.cc:514-578) with
// Mm::DensityClustering<u8, s32> (Mm/DensityClustering.tcc) over the quantised means of the mixture entries.
// lane = mixture entry: first closest cluster (s32 sum of squared differences, strict '<')
__global__ __launch_bounds__(256) void simd_presel_assign_kernel(const unsigned char* __restrict__ qmean, const uint32_t* __restrict__ k_dens,
                                                                int nk, int dim, const unsigned char* __restrict__ cm, int n_clusters,
                                                                uint32_t* __restrict__ cluster_of) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nk)
        return;
    const unsigned char* q  = qmean + (size_t)k_dens[k] * dim;
    int                  bd = INT_MAX;
    uint32_t             bc = 0;
    for (int c = 0; c < n_clusters; ++c) {
        const unsigned char* m = cm + (size_t)c * dim;  // wave-uniform
        int                  s = 0;
        for (int i = 0; i < dim; ++i) {
            const int df = (int)m[i] - (int)q[i];
            s += df * df;
        }
        if (s < bd) {
            bd = s;
            bc = (uint32_t)c;
        }
    }
    cluster_of[k] = bc;
}

// lane = frame: quantised feature (LDS, [dim][64] bytes), s32 distance to every cluster mean -> scratch [n_clusters x Tpad], the
// n_select smallest (distance, cluster) pairs are active (selectClusters sorts by distance only and leaves ties unspecified; here
// the cluster index breaks them), one 64-bit lane mask per (wave of 64 frames, cluster)
__global__ __launch_bounds__(64) void simd_presel_select_kernel(const float* __restrict__ feats, const float* __restrict__ isr, int T, int Tpad,
                                                               int dim, const unsigned char* __restrict__ cm, int n_clusters, int n_select,
                                                               int* __restrict__ g_dist, unsigned long long* __restrict__ g_masks) {
    extern __shared__ unsigned char s_q[];  // [dim][64]
    const int  lane = threadIdx.x, t = blockIdx.x * 64 + lane;
    const bool live = t < T;
    const float* x  = feats + (size_t)(live ? t : T - 1) * dim;
    for (int i = 0; i < dim; ++i)
        s_q[i * 64 + lane] = (unsigned char)simd_quantize(x[i] * isr[i]);
    for (int c = 0; c < n_clusters; ++c) {
        const unsigned char* m = cm + (size_t)c * dim;
        int                  s = 0;
        for (int i = 0; i < dim; ++i) {
            const int df = (int)s_q[i * 64 + lane] - (int)m[i];
            s += df * df;
        }
        g_dist[(size_t)c * Tpad + t] = s;
    }
    int pd = -1, pc = -1;  // distances are >= 0
    for (int sel = 0; sel < n_select; ++sel) {
        int  bd = INT_MAX, bc = n_clusters;
        bool any = false;
        for (int c = 0; c < n_clusters; ++c) {
            const int  d     = g_dist[(size_t)c * Tpad + t];
            const bool above = d > pd || (d == pd && c > pc);
            const bool lower = d < bd || (d == bd && c < bc);
            if (above && lower) {
                bd  = d;
                bc  = c;
                any = true;
            }
        }
        if (!any)
            break;
        pd = bd;
        pc = bc;
    }
    for (int c = 0; c < n_clusters; ++c) {
        const int                d      = g_dist[(size_t)c * Tpad + t];
        const bool               active = live && (d < pd || (d == pd && c <= pc));
        const unsigned long long m      = __ballot(active);
        if (lane == 0)
            g_masks[(size_t)blockIdx.x * n_clusters + c] = m;
    }
}

// simd_combine_kernel restricted to the densities of active clusters; a mixture without one keeps INT_MAX (the int class has
// no back-off score: (f32)INT_MAX / scale_)
__global__ __launch_bounds__(256) void simd_presel_combine_kernel(const int* __restrict__ dist, const int* __restrict__ cst,
                                                                 const uint32_t* __restrict__ mix_off, const uint32_t* __restrict__ k_dens,
                                                                 const uint32_t* __restrict__ cluster_of,
                                                                 const unsigned long long* __restrict__ g_masks, int n_clusters,
                                                                 float* __restrict__ scores, int T, int ld, int n_mix, int mix_tile,
                                                                 SimdScale scale) {
    const int  t    = blockIdx.y * 256 + threadIdx.x;
    const int  lane = threadIdx.x & 63;
    const bool live = t < T;
    const int  tt   = live ? t : T - 1;
    if ((t & ~63) >= T)
        return;
    const unsigned long long* masks = g_masks + (size_t)(t >> 6) * n_clusters;
    const int m0 = blockIdx.x * mix_tile, m1 = min(m0 + mix_tile, n_mix);
    for (int m = m0; m < m1; ++m) {
        const uint32_t k0 = mix_off[m], k1 = mix_off[m + 1];
        int            mn = INT_MAX;
        for (uint32_t k = k0; k < k1; ++k) {
            const unsigned long long am = masks[cluster_of[k]];  // wave-uniform
            if (am == 0ull)
                continue;
            const int s = cst[k] + dist[(size_t)k_dens[k] * ld + tt];
            if (((am >> lane) & 1ull) && s < mn)
                mn = s;
        }
        if (live)
            scores[(size_t)t * n_mix + m] = simd_score(mn, scale);
    }
}

struct GmmSimd {
    int      dim = 0, n_mix = 0, n_dens = 0, n_cov = 0;
    size_t   nk = 0;
    float    scaling = 0.f, scaling2 = 0.f;
    float*   d_isr = nullptr;             // [n_cov x dim], scaled
    unsigned char* d_qmean = nullptr;     // [n_dens x dim]
    int*     d_cst[2] = {nullptr, nullptr};  // [nk] per-density constants: SIMD-diagonal-maximum, batch-diagonal-maximum-int
    uint32_t *d_mix_off = nullptr, *d_k_dens = nullptr, *d_d_cov = nullptr;
    int*     d_dist = nullptr;
    size_t   dist_cap = 0;
    int      n_tiles_r = 0;
    int8_t*  d_A = nullptr;
    int*     d_key[2] = {nullptr, nullptr};
    bool     has_int = false;  // batch-int tables exist (pooled covariance)
    bool     mfma_v[2] = {false, false};
    float    int_scale = 0.f;
    int8_t*  d_X = nullptr;
    int*     d_nx = nullptr;
    int      cap_T = 0;
    // preselection-batch-int
    std::vector<unsigned char> h_qmean;
    std::vector<uint32_t>      h_k_dens;
    int                        ps_clusters = 0, ps_select = 0, ps_iterations = -1;  // parameters of the clustering below
    std::vector<unsigned char> h_cm;
    std::vector<uint32_t>      h_cluster_of;
    unsigned char*             d_cm = nullptr;
    uint32_t*                  d_cluster_of = nullptr;
    int*                       d_cdist = nullptr;
    unsigned long long*        d_masks = nullptr;
    int                        ps_cap_T = 0;
};

template<class T>
static int upload(T** dst, const T* src, size_t n) {
    AMX_HIP(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
    if (n)
        AMX_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return AMX_OK;
}

static int host_quantize(float v) {
    const float r  = std::round(v);
    const int   qi = (r >= -2147483648.f && r < 2147483648.f) ? (int)r : INT_MIN;
    const int   q  = (int)((unsigned)qi + 128u);
    return std::min(std::max(q, 0), 255);
}

}  // namespace amx

extern "C" {

void amx_internal_gmm_simd_destroy(void* p) {
    amx::GmmSimd* s = (amx::GmmSimd*)p;
    if (!s)
        return;
    hipFree(s->d_isr);
    hipFree(s->d_qmean);
    hipFree(s->d_cst[0]);
    hipFree(s->d_cst[1]);
    hipFree(s->d_mix_off);
    hipFree(s->d_k_dens);
    hipFree(s->d_d_cov);
    hipFree(s->d_dist);
    hipFree(s->d_A);
    hipFree(s->d_key[0]);
    hipFree(s->d_key[1]);
    hipFree(s->d_X);
    hipFree(s->d_nx);
    hipFree(s->d_cm);
    hipFree(s->d_cluster_of);
    hipFree(s->d_cdist);
    hipFree(s->d_masks);
    delete s;
}

// float / double -> s32 as the reference's x86-64 build converts (cvttss2si / cvttsd2si): truncation, and the "integer indefinite"
// value INT_MIN for NaN and anything out of range -- a zero-weight density (log weight -DBL_MAX, e.g. from a version < 2.0 .pms
// file) makes the constant +inf.  A plain C++ cast is undefined behaviour there.
static inline int cvt_s32_x86(double v) {
    return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (-2147483647 - 1);
}

// SimdGaussDiagonalMaximumFeatureScorer::init + buildMixtureTable (Mm/SimdFeatureScorer.cc:68-137)
int amx_internal_gmm_simd_create(const amx_gmm_model* m, int contract_fma, void** out, float* scaling_out) {
    using namespace amx;
    *out = nullptr;
    const int    dim = m->dim;
    const size_t nk  = m->mix_offsets[m->n_mix];
    std::vector<float> isr((size_t)m->n_cov * dim), lognorm(m->n_cov);
    for (int c = 0; c < m->n_cov; ++c) {
        double lsum = 0;
        for (int i = 0; i < dim; ++i) {
            const float v            = m->variances[(size_t)c * dim + i];
            isr[(size_t)c * dim + i] = (float)1 / (float)std::sqrt((double)v);
            lsum += std::log((double)std::fabs(v));
        }
        // gaussLogNormFactor (Mm/Utilities.hh:70-75): N * log(2 pi) + logNorm is one vfmadd in the reference's default build.  The
        // scorers' other arithmetic is integer; the batch-int constant (s32)(logNorm scale^2 - scale_ logWeight), an f64 expression of two
        // products, was not examined under the default flags and is evaluated unfused in both modes (the class is parity unpinned).
        lognorm[c] = (float)(contract_fma ? std::fma((double)dim, std::log((double)2 * M_PI), lsum) : (double)dim * std::log((double)2 * M_PI) + lsum);
    }
    float min_mean = FLT_MAX, max_mean = -FLT_MAX;  // getScaling (:112-130): over all densities, unscaled 1/sigma
    for (int d = 0; d < m->n_dens; ++d) {
        const float* mu = m->means + (size_t)m->dens_mean[d] * dim;
        const float* is = isr.data() + (size_t)m->dens_cov[d] * dim;
        for (int i = 0; i < dim; ++i) {
            const float dm = mu[i] * is[i];
            min_mean       = std::min(min_mean, dm);
            max_mean       = std::max(max_mean, dm);
        }
    }
    const float interval = 2 * std::max(std::fabs(min_mean), std::fabs(max_mean));
    const float scaling  = (float)((float)255 / (1.25 * interval));  // quantizationScalingFactor (:132-137)
    const float scaling2 = scaling * scaling;
    if (scaling_out)
        *scaling_out = scaling;
    for (int c = 0; c < m->n_cov; ++c) {  // CovarianceFeatureScorerElement::scale
        for (int i = 0; i < dim; ++i)
            isr[(size_t)c * dim + i] = isr[(size_t)c * dim + i] * scaling;
        lognorm[c] = lognorm[c] * (scaling * scaling);
    }
    std::vector<unsigned char> qmean((size_t)m->n_dens * dim);
    for (int d = 0; d < m->n_dens; ++d)
        for (int i = 0; i < dim; ++i)
            qmean[(size_t)d * dim + i] =
                    (unsigned char)host_quantize(m->means[(size_t)m->dens_mean[d] * dim + i] * isr[(size_t)m->dens_cov[d] * dim + i]);
    std::vector<int> cst(nk);
    for (size_t k = 0; k < nk; ++k) {  // buildMixtureTable (:92-100) + createDensityElement (Mm/IntelOptimization.cc:37-46)
        const double scaled  = (double)(scaling2 * -2) * m->log_weight[k];
        const float  asScore = (float)scaled;
        cst[k]               = cvt_s32_x86((double)(asScore + lognorm[m->dens_cov[m->dens_index[k]]]));
    }
    // Mm::BatchIntFeatureScorer::init (Mm/BatchFeatureScorer.cc:375-417), pooled covariance only: the same quantised means
    // (quantizationScale is getScaling's formula), scale_ = (f32)(2.0 * scale^2), constant = (s32)(logNorm * scale^2 - scale_ * logWeight)
    // with the subtraction in f64
    std::vector<int> cst_int;
    const float      int_scale = (float)(2.0 * scaling2);
    if (m->n_cov == 1) {
        cst_int.resize(nk);
        const float log_norm_factor = lognorm[0];  // already logNormalizationFactor() * scaleSquared
        for (size_t k = 0; k < nk; ++k)
            cst_int[k] = cvt_s32_x86((double)log_norm_factor - (double)int_scale * m->log_weight[k]);
    }
    GmmSimd* s  = new GmmSimd;
    s->dim      = dim;
    s->n_mix    = m->n_mix;
    s->n_dens   = m->n_dens;
    s->n_cov    = m->n_cov;
    s->nk       = nk;
    s->scaling  = scaling;
    s->scaling2 = scaling2;
    s->has_int  = m->n_cov == 1;
    s->int_scale = int_scale;
    if (s->has_int) {
        s->h_qmean = qmean;
        s->h_k_dens.assign(m->dens_index, m->dens_index + nk);
    }
    int r;
    if ((r = upload(&s->d_isr, isr.data(), isr.size())) != AMX_OK || (r = upload(&s->d_qmean, qmean.data(), qmean.size())) != AMX_OK ||
        (r = upload(&s->d_cst[0], cst.data(), cst.size())) != AMX_OK ||
        (s->has_int && (r = upload(&s->d_cst[1], cst_int.data(), cst_int.size())) != AMX_OK) ||
        (r = upload(&s->d_mix_off, m->mix_offsets, (size_t)m->n_mix + 1)) != AMX_OK ||
        (r = upload(&s->d_k_dens, m->dens_index, nk)) != AMX_OK || (r = upload(&s->d_d_cov, m->dens_cov, (size_t)m->n_dens)) != AMX_OK) {
        amx_internal_gmm_simd_destroy(s);
        return r;
    }
    // MFMA path tables: pooled covariance, <= 16 densities per mixture, one 64-byte row per slot, keys that fit 28 bits
    uint32_t kmax = 0;
    for (int i = 0; i < m->n_mix; ++i)
        kmax = std::max(kmax, m->mix_offsets[i + 1] - m->mix_offsets[i]);
    amx::Tuning tune;
    if (!tune.parse(m->tuning, amx::gmm_tuning_keys, "amx_gmm_create")) {
        amx_internal_gmm_simd_destroy(s);
        return AMX_ERR_INVALID;
    }
    int want_i = 1;
    if (!tune.get_int("simd_mfma", 1, 0, 1, &want_i, "amx_gmm_create")) {
        amx_internal_gmm_simd_destroy(s);
        return AMX_ERR_INVALID;
    }
    const bool want = want_i != 0;
    if (want && m->n_cov == 1 && dim <= 64 && kmax <= 16) {
        const int           n_tiles = (m->n_mix + 15) / 16;
        std::vector<int8_t> A((size_t)n_tiles * 256 * 64, 0);
        for (int v = 0; v < 2; ++v) {
            if (v == 1 && !s->has_int)
                break;
            const std::vector<int>& cv = v ? cst_int : cst;
            std::vector<int>        key((size_t)n_tiles * 256, INT_MIN);  // negated keys; INT_MIN marks an empty slot
            bool                    fits = true;
            for (int i = 0; i < m->n_mix && fits; ++i)
                for (uint32_t k = m->mix_offsets[i]; k < m->mix_offsets[i + 1]; ++k) {
                    const uint32_t slot = k - m->mix_offsets[i];
                    const int      tile = i >> 4, ml = i & 15;
                    const size_t   row  = (size_t)tile * 256 + (ml >> 1) * 32 + (slot >> 2) * 8 + (ml & 1) * 4 + (slot & 3);
                    const unsigned char* q = qmean.data() + (size_t)m->dens_index[k] * dim;
                    long long      na = 0;
                    for (int x = 0; x < dim; ++x) {
                        const int a     = (int)q[x] - 128;
                        A[row * 64 + x] = (int8_t)a;
                        na += a * a;
                    }
                    const long long base = (long long)cv[k] + na;
                    if (base >= (1ll << 27) - (1ll << 22) || base <= -(1ll << 27) + (1ll << 22))
                        fits = false;  // key - 32 a'.b' must stay inside 32 bits with room for the slot number
                    key[(size_t)tile * 256 + ml * 16 + slot] = -(int)(base * 16 + slot);
                }
            if (fits) {
                if ((!s->d_A && (r = upload(&s->d_A, A.data(), A.size())) != AMX_OK) ||
                    (r = upload(&s->d_key[v], key.data(), key.size())) != AMX_OK) {
                    amx_internal_gmm_simd_destroy(s);
                    return r;
                }
                s->mfma_v[v] = true;
                s->n_tiles_r = n_tiles;
            }
        }
    }
    *out = s;
    return AMX_OK;
}

float amx_internal_gmm_simd_scaling(const void* p) {
    return p ? ((const amx::GmmSimd*)p)->scaling : 0.f;
}

int amx_internal_gmm_simd_score(void* p, amx_ctx* ctx, int variant, const float* feats_dev, int T, float* scores_dev, uint32_t* best_dev) {
    using namespace amx;
    GmmSimd*    s  = (GmmSimd*)p;
    hipStream_t st = ctx->stream;
    SimdScale   scale;
    scale.b = 2.0 * (double)s->scaling2;
    scale.y = 1.0 / scale.b;
    scale.int_scale = variant ? s->int_scale : 0.f;
    if (variant && !s->has_int) {  // BatchIntFeatureScorer::init: criticalError
        amx::set_error("amx_gmm_score_dev: feature scorer supports only globally pooled variance");
        return AMX_ERR_INVALID;
    }
    if (s->mfma_v[variant]) {
        const int chunk = 65536;
        for (int t0 = 0; t0 < T; t0 += chunk) {
            const int Tc = std::min(chunk, T - t0), Tpad = (Tc + 255) / 256 * 256;
            if (Tpad > s->cap_T) {
                hipFree(s->d_X);
                hipFree(s->d_nx);
                s->d_X   = nullptr;
                s->d_nx  = nullptr;
                s->cap_T = 0;
                AMX_HIP(hipMalloc((void**)&s->d_X, (size_t)Tpad * 64));
                AMX_HIP(hipMalloc((void**)&s->d_nx, (size_t)Tpad * 4));
                s->cap_T = Tpad;
            }
            const float* x = feats_dev + (size_t)t0 * s->dim;
            {
                ScopedKernelTimer timer(ctx, "gmm_simd_quantize");
                hipLaunchKernelGGL(simd_quantize_kernel, dim3(Tpad / 4), dim3(256), 0, st, x, s->d_isr, s->d_X, s->d_nx, Tc, Tpad, s->dim);
            }
            const int tiles_t = Tpad / 256;
            int       r_split = 1;
            while (tiles_t * r_split < 512 && r_split * 2 <= s->n_tiles_r)
                r_split *= 2;
            static bool attr = false;
            if (!attr) {
                AMX_HIP(hipFuncSetAttribute((const void*)simd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSimdLds));
                attr = true;
            }
            ScopedKernelTimer timer(ctx, "gmm_simd");
            hipLaunchKernelGGL(simd_mfma_kernel, dim3(tiles_t * r_split), dim3(512), kSimdLds, st, s->d_A, s->d_X, s->d_key[variant], s->d_nx,
                               scores_dev + (size_t)t0 * s->n_mix, best_dev ? best_dev + (size_t)t0 * s->n_mix : nullptr, Tc, s->n_mix,
                               s->n_tiles_r, r_split, scale);
        }
        AMX_HIP(hipGetLastError());
        return AMX_OK;
    }
    // general path: distance scratch bounded at 256 MB
    int chunk = (int)std::min<size_t>((size_t)T, std::max<size_t>(256, ((size_t)64 << 20) / (size_t)s->n_dens / 256 * 256));
    chunk     = (chunk + 255) / 256 * 256;
    const size_t need = (size_t)s->n_dens * chunk;
    if (need > s->dist_cap) {
        hipFree(s->d_dist);
        s->d_dist   = nullptr;
        s->dist_cap = 0;
        AMX_HIP(hipMalloc((void**)&s->d_dist, need * 4));
        s->dist_cap = need;
    }
    for (int t0 = 0; t0 < T; t0 += chunk) {
        const int    Tc = std::min(chunk, T - t0), fblocks = (Tc + 255) / 256;
        const float* x  = feats_dev + (size_t)t0 * s->dim;
        int          dt = 64;
        while (dt > 4 && (long)((s->n_dens + dt - 1) / dt) * fblocks < 1024)
            dt /= 2;
        {
            ScopedKernelTimer timer(ctx, "gmm_simd_dist");
            hipLaunchKernelGGL(simd_dist_kernel, dim3((s->n_dens + dt - 1) / dt, fblocks), dim3(256), 0, st, x, s->d_isr, s->d_qmean, s->d_d_cov,
                               s->d_dist, Tc, chunk, s->dim, s->n_dens, dt);
        }
        int mt = 16;
        while (mt > 1 && (long)((s->n_mix + mt - 1) / mt) * fblocks < 1024)
            mt /= 2;
        ScopedKernelTimer timer(ctx, "gmm_simd");
        hipLaunchKernelGGL(simd_combine_kernel, dim3((s->n_mix + mt - 1) / mt, fblocks), dim3(256), 0, st, s->d_dist, s->d_cst[variant], s->d_mix_off,
                           s->d_k_dens, scores_dev + (size_t)t0 * s->n_mix, best_dev ? best_dev + (size_t)t0 * s->n_mix : nullptr, Tc, chunk,
                           s->n_mix, mt, scale);
    }
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

// glibc's srand(1) / rand() stream (TYPE_3 additive feedback generator), as in gmm_presel.hip: initializeClusters draws from it
static void simd_glibc_rand_init(std::vector<int32_t>& st) {
    int32_t r[34];
    r[0] = 1;
    for (int i = 1; i < 31; ++i) {
        const int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        int64_t       w  = 16807 * lo - 2836 * hi;
        if (w < 0)
            w += 2147483647;
        r[i] = (int32_t)w;
    }
    st.assign(r, r + 31);
    for (int i = 31; i < 34; ++i)
        st.push_back(st[i - 31]);
    for (int i = 34; i < 344; ++i)
        st.push_back((int32_t)((uint32_t)st[i - 31] + (uint32_t)st[i - 3]));
}
static int simd_glibc_rand_next(std::vector<int32_t>& st) {
    const size_t   i = st.size();
    const uint32_t v = (uint32_t)st[i - 31] + (uint32_t)st[i - 3];
    st.push_back((int32_t)v);
    if (st.size() > 4096)
        st.erase(st.begin(), st.end() - 64);
    return (int)(v >> 1);
}

// DensityClustering<u8, s32>::build over the quantised means of the mixture entries (rebuilt when the parameters change)
int amx_internal_gmm_simd_presel_build(void* p, amx_ctx* ctx, int n_clusters, int n_select, int iterations) {
    using namespace amx;
    GmmSimd* s = (GmmSimd*)p;
    if (!s->has_int) {
        amx::set_error("amx_gmm_score_dev: feature scorer supports only globally pooled variance");
        return AMX_ERR_INVALID;
    }
    const size_t nk = s->nk;
    if ((size_t)n_clusters > nk)
        n_clusters = (int)nk;  // "reducing number of clusters ... because there are too few densities"
    AMX_REQUIRE(n_clusters >= 1 && n_clusters <= 256, AMX_ERR_INVALID, "preselection: clusters must be in 1..256 (got %d)", n_clusters);
    AMX_REQUIRE(n_select >= 1 && n_select <= n_clusters, AMX_ERR_INVALID, "preselection: select-clusters (%d) must be in 1..clusters (%d)", n_select,
                n_clusters);
    if (s->d_cm && s->ps_clusters == n_clusters && s->ps_iterations == iterations) {
        s->ps_select = n_select;
        return AMX_OK;
    }
    const int dim = s->dim;
    s->h_cm.assign((size_t)n_clusters * dim, 0);
    s->h_cluster_of.assign(nk, 0u);
    {
        std::vector<int32_t> st;
        simd_glibc_rand_init(st);
        std::vector<char> used(nk, 0);
        for (int c = 0; c < n_clusters; ++c) {
            uint32_t pick;
            do {
                pick = (uint32_t)simd_glibc_rand_next(st) % (uint32_t)nk;
            } while (used[pick]);
            used[pick] = 1;
            memcpy(&s->h_cm[(size_t)c * dim], &s->h_qmean[(size_t)s->h_k_dens[pick] * dim], (size_t)dim);
        }
    }
    AMX_HIP(hipSetDevice(ctx->device));
    hipFree(s->d_cm);
    hipFree(s->d_cluster_of);
    s->d_cm         = nullptr;
    s->d_cluster_of = nullptr;
    AMX_HIP(hipMalloc((void**)&s->d_cm, s->h_cm.size()));
    AMX_HIP(hipMalloc((void**)&s->d_cluster_of, std::max<size_t>(nk, 1) * 4));
    std::vector<double> sums((size_t)n_clusters * dim);
    std::vector<size_t> cnt(n_clusters);
    for (int it = 0; it < iterations; ++it) {
        AMX_HIP(hipMemcpyAsync(s->d_cm, s->h_cm.data(), s->h_cm.size(), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(simd_presel_assign_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, ctx->stream, s->d_qmean, s->d_k_dens,
                           (int)nk, dim, s->d_cm, n_clusters, s->d_cluster_of);
        AMX_HIP(hipMemcpyAsync(s->h_cluster_of.data(), s->d_cluster_of, nk * 4, hipMemcpyDeviceToHost, ctx->stream));
        AMX_HIP(hipStreamSynchronize(ctx->stream));
        std::fill(sums.begin(), sums.end(), 0.0);  // updateClusterMeans: f64 sums in density order / count -> u8 (truncation)
        std::fill(cnt.begin(), cnt.end(), (size_t)0);
        for (size_t k = 0; k < nk; ++k) {
            const uint32_t       c = s->h_cluster_of[k];
            const unsigned char* q = &s->h_qmean[(size_t)s->h_k_dens[k] * dim];
            double*              sm = &sums[(size_t)c * dim];
            for (int i = 0; i < dim; ++i)
                sm[i] = sm[i] + (double)q[i];
            ++cnt[c];
        }
        for (int c = 0; c < n_clusters; ++c)
            if (cnt[c])
                for (int i = 0; i < dim; ++i)
                    s->h_cm[(size_t)c * dim + i] = (unsigned char)(sums[(size_t)c * dim + i] / (double)cnt[c]);
    }
    AMX_HIP(hipMemcpy(s->d_cm, s->h_cm.data(), s->h_cm.size(), hipMemcpyHostToDevice));
    AMX_HIP(hipMemcpy(s->d_cluster_of, s->h_cluster_of.data(), nk * 4, hipMemcpyHostToDevice));
    s->ps_clusters   = n_clusters;
    s->ps_select     = n_select;
    s->ps_iterations = iterations;
    return AMX_OK;
}

int amx_internal_gmm_simd_presel_info(const void* p, int* n_clusters, uint32_t* cluster_of, float* cluster_means) {
    const amx::GmmSimd* s = (const amx::GmmSimd*)p;
    if (n_clusters)
        *n_clusters = s->ps_clusters;
    if (cluster_of)
        memcpy(cluster_of, s->h_cluster_of.data(), s->nk * 4);
    if (cluster_means)
        for (size_t i = 0; i < s->h_cm.size(); ++i)
            cluster_means[i] = (float)s->h_cm[i];
    return AMX_OK;
}

int amx_internal_gmm_simd_presel_score(void* p, amx_ctx* ctx, const float* feats_dev, int T, float* scores_dev) {
    using namespace amx;
    GmmSimd*    s  = (GmmSimd*)p;
    hipStream_t st = ctx->stream;
    SimdScale   scale;
    scale.b         = 2.0 * (double)s->scaling2;
    scale.y         = 1.0 / scale.b;
    scale.int_scale = s->int_scale;
    int chunk = (int)std::min<size_t>((size_t)T, std::max<size_t>(256, ((size_t)64 << 20) / (size_t)s->n_dens / 256 * 256));
    chunk     = (chunk + 255) / 256 * 256;
    const size_t need = (size_t)s->n_dens * chunk;
    if (need > s->dist_cap) {
        hipFree(s->d_dist);
        s->d_dist   = nullptr;
        s->dist_cap = 0;
        AMX_HIP(hipMalloc((void**)&s->d_dist, need * 4));
        s->dist_cap = need;
    }
    if (chunk > s->ps_cap_T) {
        hipFree(s->d_cdist);
        hipFree(s->d_masks);
        s->d_cdist  = nullptr;
        s->d_masks  = nullptr;
        s->ps_cap_T = 0;
        AMX_HIP(hipMalloc((void**)&s->d_cdist, (size_t)256 * chunk * 4));
        AMX_HIP(hipMalloc((void**)&s->d_masks, (size_t)(chunk / 64) * 256 * 8));
        s->ps_cap_T = chunk;
    }
    for (int t0 = 0; t0 < T; t0 += chunk) {
        const int    Tc = std::min(chunk, T - t0), fblocks = (Tc + 255) / 256;
        const float* x  = feats_dev + (size_t)t0 * s->dim;
        int          dt = 64;
        while (dt > 4 && (long)((s->n_dens + dt - 1) / dt) * fblocks < 1024)
            dt /= 2;
        {
            ScopedKernelTimer timer(ctx, "gmm_simd_dist");
            hipLaunchKernelGGL(simd_dist_kernel, dim3((s->n_dens + dt - 1) / dt, fblocks), dim3(256), 0, st, x, s->d_isr, s->d_qmean, s->d_d_cov,
                               s->d_dist, Tc, chunk, s->dim, s->n_dens, dt);
        }
        ScopedKernelTimer timer(ctx, "gmm_simd");
        hipLaunchKernelGGL(simd_presel_select_kernel, dim3((Tc + 63) / 64), dim3(64), (size_t)s->dim * 64, st, x, s->d_isr, Tc, chunk, s->dim,
                           s->d_cm, s->ps_clusters, s->ps_select, s->d_cdist, s->d_masks);
        int mt = 16;
        while (mt > 1 && (long)((s->n_mix + mt - 1) / mt) * fblocks < 1024)
            mt /= 2;
        hipLaunchKernelGGL(simd_presel_combine_kernel, dim3((s->n_mix + mt - 1) / mt, fblocks), dim3(256), 0, st, s->d_dist, s->d_cst[1],
                           s->d_mix_off, s->d_k_dens, s->d_cluster_of, s->d_masks, s->ps_clusters, scores_dev + (size_t)t0 * s->n_mix, Tc, chunk,
                           s->n_mix, mt, scale);
    }
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

}  // extern "C"
