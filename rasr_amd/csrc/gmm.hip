// gmm.hip -- diagonal-covariance GMM emission scorer for gfx950 and the amx_gmm_* ABI.
//
// Replaces Mm::GaussDiagonalMaximumFeatureScorer / GaussDiagonalSumFeatureScorer
// (Mm/GaussDiagonalMaximumFeatureScorer.cc:116-298) evaluated for ALL emissions of a batch of
// frames (the reference evaluates lazily per active state; a batch scorer answers score(e)
// from the [T x M] matrix like Nn::BatchFeatureScorer::getScore does).
//
// Mapping to the hardware
//   * lane = frame.  A wavefront owns 64 consecutive frames and keeps each frame's feature
//     vector in VGPRs (DIM registers per lane).  The model is wave-uniform: means, 1/sigma and
//     the per-density constants stream through the SCALAR data path (s_load via the constant
//     cache), so every VALU instruction takes one SGPR model operand and one VGPR feature
//     operand -- no LDS, no cross-lane traffic, and the reduction over the densities of a
//     mixture is a sequential per-lane loop exactly like the reference's.
//   * direct kernel: workgroup = 4 frame-waves x one tile of MT mixtures (densities private to
//     mixtures, CART-style models).
//   * tied models (sum K_m >> #densities): stage 1 computes every density's distance once
//     (dist[d][t], coalesced along t), stage 2 combines dist with the mixture weights.
//
// Exactness (max mode): the squared distance is accumulated in the reference's SSE order --
// four strided partial sums over dim&~3, (l0+l1)+(l2+l3), scalar tail -- with separate
// multiply and add (-ffp-contract=off) -- or, with amx_gmm_model.tuning contract=fma, with the accumulate `sum += df * df` as ONE
// fused multiply-add, which is what the reference's DEFAULT build (-march=native, GCC's -ffp-contract=fast) executes on an FMA host
// (template parameter FMA of every kernel that evaluates the distance; sq_acc in gmm_device.hpp); the combine
// (f64)m2lw + (f64)logNorm + (f64)dist is done in f64 (the first f64 addition is folded on the host, it does not depend on the
// frame); the running best is an f32 compared as f64 with strict '>', so the first minimum wins.  Scores and best-density indices
// are therefore bit-identical to the reference built with -DMARCH=x86-64 (contract=off, the default) or to its default build
// (contract=fma) -- INTEGRATION.md has the table.
#include "common.hpp"
#include "gmm_device.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>

namespace amx {

struct GmmParams {
    const float* __restrict__ feats;     // [T x dim]
    float* __restrict__ scores;          // [T x n_mix]
    uint32_t* __restrict__ best;         // nullable [T x n_mix]
    const uint32_t* __restrict__ mix_off;  // [n_mix+1]
    const uint32_t* __restrict__ k_mean;   // [nk] mean row of entry k
    const uint32_t* __restrict__ k_cov;    // [nk] covariance row of entry k
    const double* __restrict__ k_c64;      // [nk] (f64)m2lw + (f64)logNorm
    const float* __restrict__ k_c32;       // [nk] m2lw + logNorm in f32 (sum mode)
    const float* __restrict__ means;       // [n_mean x dim]
    const float* __restrict__ isr;         // [n_cov x dim]
    int T, dim, n_mix, mix_tile;
};

// distance in the reference's association order; mu / is are wave-uniform pointers
template<int DIM, bool FMA = false>
__device__ __forceinline__ float gmm_distance(const float (&x)[DIM], const float* __restrict__ mu, const float* __restrict__ is) {
    float         l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
    constexpr int EFF = DIM & ~3;
#pragma unroll
    for (int i = 0; i < EFF; i += 4) {
        float d0 = (mu[i] - x[i]) * is[i];
        float d1 = (mu[i + 1] - x[i + 1]) * is[i + 1];
        float d2 = (mu[i + 2] - x[i + 2]) * is[i + 2];
        float d3 = (mu[i + 3] - x[i + 3]) * is[i + 3];
        l0       = sq_acc<FMA>(d0, l0);
        l1       = sq_acc<FMA>(d1, l1);
        l2       = sq_acc<FMA>(d2, l2);
        l3       = sq_acc<FMA>(d3, l3);
    }
    float result = 0.f;
    result       = result + ((l0 + l1) + (l2 + l3));
#pragma unroll
    for (int i = EFF; i < DIM; ++i) {
        float df = (mu[i] - x[i]) * is[i];
        result   = sq_acc<FMA>(df, result);
    }
    return result;
}

// runtime-dimension variant: features live in LDS as [dim][64] (one column per lane)
template<bool FMA = false>
__device__ __forceinline__ float gmm_distance_rt(const float* xs, int dim, const float* __restrict__ mu, const float* __restrict__ is) {
    float     l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
    const int eff = dim & ~3;
    for (int i = 0; i < eff; i += 4) {
        float d0 = (mu[i] - xs[i * 64]) * is[i];
        float d1 = (mu[i + 1] - xs[(i + 1) * 64]) * is[i + 1];
        float d2 = (mu[i + 2] - xs[(i + 2) * 64]) * is[i + 2];
        float d3 = (mu[i + 3] - xs[(i + 3) * 64]) * is[i + 3];
        l0       = sq_acc<FMA>(d0, l0);
        l1       = sq_acc<FMA>(d1, l1);
        l2       = sq_acc<FMA>(d2, l2);
        l3       = sq_acc<FMA>(d3, l3);
    }
    float result = 0.f;
    result       = result + ((l0 + l1) + (l2 + l3));
    for (int i = eff; i < dim; ++i) {
        float df = (mu[i] - xs[i * 64]) * is[i];
        result   = sq_acc<FMA>(df, result);
    }
    return result;
}

struct MaxState {
    float    best   = FLT_MAX;
    double   best_d = (double)FLT_MAX;
    uint32_t idx    = 0xffffffffu;
    __device__ __forceinline__ void add(double c64, float, float dist, uint32_t k) {
        double s = c64 + (double)dist;
        if (best_d > s) {
            best   = (float)s;
            best_d = (double)best;
            idx    = k;
        }
    }
    // distance already widened to f64 (uniform-list path keeps no f32 copy of the running best)
    __device__ __forceinline__ void add_d(double c64, double dist64, uint32_t k) {
        const double s  = c64 + dist64;
        const float  sf = (float)s;
        const bool   c  = best_d > s;
        best            = c ? sf : best;
        idx             = c ? k : idx;
        best_d          = (double)best;
    }
    __device__ __forceinline__ float result_d() const { return 0.5f * best; }
    __device__ __forceinline__ float result() const { return 0.5f * best; }
    static constexpr bool kF64 = true;
};

struct SumState {
    float    best = FLT_MAX;
    float    sum  = 0.f;
    uint32_t idx  = 0xffffffffu;
    __device__ __forceinline__ void add(double, float c32, float dist, uint32_t k) {
        float score = c32 + dist;  // (m2lw + logNorm) + dist, all f32
        float s     = 0.5f * score;
        if (best > s) {
            sum  = sum * expf(s - best) + 1.f;
            best = s;
            idx  = k;
        }
        else
            sum = sum + expf(best - s);
    }
    __device__ __forceinline__ float result() const { return best - logf(sum); }
    __device__ __forceinline__ void  add_d(double, double, uint32_t) {}
    __device__ __forceinline__ float result_d() const { return result(); }
    static constexpr bool kF64 = false;
};

// DIM > 0: features in registers; DIM == 0: runtime dimension, features in LDS
struct GmmDims {
    int T, dim, n_mix, mix_tile;
};

// All model tables are separate `const __restrict__` kernel arguments so that the compiler may
// prove them read-only and fetch them through the scalar cache (s_load) -- struct members lose
// the qualifier and fall back to per-lane vector loads.
template<int DIM, class State, bool FMA = false>
__global__ __launch_bounds__(256) void gmm_direct_kernel(const float* __restrict__ g_feats, float* __restrict__ g_scores,
                                                        uint32_t* __restrict__ g_best, const uint32_t* __restrict__ g_mix_off,
                                                        const uint32_t* __restrict__ g_k_mean, const uint32_t* __restrict__ g_k_cov,
                                                        const double* __restrict__ g_k_c64, const float* __restrict__ g_k_c32,
                                                        const float* __restrict__ g_means, const float* __restrict__ g_isr,
                                                        GmmDims dims) {
    struct {
        const float* __restrict__ feats; float* __restrict__ scores; uint32_t* __restrict__ best;
        const uint32_t* __restrict__ mix_off; const uint32_t* __restrict__ k_mean; const uint32_t* __restrict__ k_cov;
        const double* __restrict__ k_c64; const float* __restrict__ k_c32; const float* __restrict__ means;
        const float* __restrict__ isr; int T, dim, n_mix, mix_tile;
    } p = {g_feats, g_scores, g_best, g_mix_off, g_k_mean, g_k_cov, g_k_c64, g_k_c32, g_means, g_isr,
           dims.T, dims.dim, dims.n_mix, dims.mix_tile};
    extern __shared__ float xs_all[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int t    = (blockIdx.y * 4 + wave) * 64 + lane;
    const bool live = t < p.T;
    const int  tt   = live ? t : (p.T - 1);

    float  x[DIM > 0 ? DIM : 1];
    float* xs = nullptr;
    if (DIM > 0) {
#pragma unroll
        for (int i = 0; i < DIM; ++i)
            x[i] = p.feats[(size_t)tt * DIM + i];
    }
    else {
        xs = xs_all + wave * 64 * p.dim + lane;
        for (int i = 0; i < p.dim; ++i)
            xs[i * 64] = p.feats[(size_t)tt * p.dim + i];
    }
    // wave-uniform loop bounds (blockIdx only)
    const int m0 = blockIdx.x * p.mix_tile;
    const int m1 = min(m0 + p.mix_tile, p.n_mix);
    for (int m = m0; m < m1; ++m) {
        const uint32_t k0 = p.mix_off[m], k1 = p.mix_off[m + 1];
        State          st;
        for (uint32_t k = k0; k < k1; ++k) {
            const float* mu = p.means + (size_t)p.k_mean[k] * (DIM > 0 ? DIM : p.dim);
            const float* is = p.isr + (size_t)p.k_cov[k] * (DIM > 0 ? DIM : p.dim);
            float        dist;
            if (DIM > 0)
                dist = gmm_distance<(DIM > 0 ? DIM : 1), FMA>(x, mu, is);
            else
                dist = gmm_distance_rt<FMA>(xs, p.dim, mu, is);
            st.add(p.k_c64[k], p.k_c32[k], dist, k - k0);
        }
        if (live) {
            p.scores[(size_t)t * p.n_mix + m] = st.result();
            if (p.best)
                p.best[(size_t)t * p.n_mix + m] = st.idx;
        }
    }
}

// ---- Mm::BatchFloatFeatureScorer ("batch-diagonal-maximum-float", Mm/BatchFeatureScorer.cc:164-254): pooled
// covariance only; means and features pre-multiplied by 1/sigma, per-density constant c = (f32)(logNorm - 2 logw);
// two 4-lane f32 accumulators over 8-wide blocks, lane 0 of the first starts at c; a = s1 + s2;
// result = (a3 + a1) + (a2 + a0); min over the densities; 0.5 * min.  3 ops per dimension instead of 4.
template<int DIM, bool FMA = false>
__global__ __launch_bounds__(256) void gmm_batch_float_kernel(const float* __restrict__ g_feats, float* __restrict__ g_scores,
                                                             const uint32_t* __restrict__ g_mix_off, const uint32_t* __restrict__ g_k_mean,
                                                             const float* __restrict__ g_k_const, const float* __restrict__ g_smeans,
                                                             const float* __restrict__ g_isr0, GmmDims dims) {
    const int  lane = threadIdx.x & 63;
    const int  wave = threadIdx.x >> 6;
    const int  t    = (blockIdx.y * 4 + wave) * 64 + lane;
    const bool live = t < dims.T;
    const int  tt   = live ? t : (dims.T - 1);
    const int  dim = DIM > 0 ? DIM : dims.dim;
    ScaledRow<DIM> x;
    x.load(g_feats + (size_t)tt * dim, g_isr0, dim);  // setFeature: f * variance_
    const int m0 = blockIdx.x * dims.mix_tile;
    const int m1 = min(m0 + dims.mix_tile, dims.n_mix);
    for (int m = m0; m < m1; ++m) {
        const uint32_t k0 = g_mix_off[m], k1 = g_mix_off[m + 1];
        float          best = FLT_MAX;
        for (uint32_t k = k0; k < k1; ++k) {
            const float* mu = g_smeans + (size_t)g_k_mean[k] * dim;
            const float  r  = batch_float_distance<DIM, FMA>(mu, x, g_k_const[k], dim);
            best           = best < r ? best : r;  // _mm_min_ps(score, r) = score < r ? score : r: a NaN sum REPLACES the score, a later finite one the NaN
                                                     // (Mm/BatchFeatureScorer.cc:245; tests/test_contract.py holds the reference's own function text to it)
        }
        if (live)
            g_scores[(size_t)t * dims.n_mix + m] = best < FLT_MAX ? 0.5f * best : best;
    }
}

// ---- AssigningContextScorer::bestDensity(e) for ONE mixture per frame (Mm/AssigningFeatureScorer.hh:127-131 ->
// GaussDiagonalMaximumFeatureScorer::calculateScoreAndDensity, Mm/GaussDiagonalMaximumFeatureScorer.cc:116-142): what the Viterbi
// accumulation asks of the scorer -- the best density of the ALIGNED mixture, not of all of them.  Thread = frame; the reference's
// arithmetic and rule (gmm_distance's operation order, MaxState), so index and score equal the full pass's entry (t, mixture[t]).
template<bool FMA>
__global__ __launch_bounds__(256) void gmm_best_density_kernel(const float* __restrict__ feats, const uint32_t* __restrict__ mixture,
                                                              uint32_t* __restrict__ best, float* __restrict__ score, const uint32_t* __restrict__ mix_off,
                                                              const uint32_t* __restrict__ k_mean, const uint32_t* __restrict__ k_cov,
                                                              const double* __restrict__ k_c64, const float* __restrict__ means,
                                                              const float* __restrict__ isr, int T, int dim, int n_mix) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T)
        return;
    const uint32_t m = mixture[t];
    MaxState       st;
    if (m < (uint32_t)n_mix) {
        const float*   x  = feats + (size_t)t * dim;
        const uint32_t k0 = mix_off[m], k1 = mix_off[m + 1];
        const int      eff = dim & ~3;
        for (uint32_t k = k0; k < k1; ++k) {
            const float* mu = means + (size_t)k_mean[k] * dim;
            const float* is = isr + (size_t)k_cov[k] * dim;
            float        l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
            for (int i = 0; i < eff; i += 4) {
                const float d0 = (mu[i] - x[i]) * is[i];
                const float d1 = (mu[i + 1] - x[i + 1]) * is[i + 1];
                const float d2 = (mu[i + 2] - x[i + 2]) * is[i + 2];
                const float d3 = (mu[i + 3] - x[i + 3]) * is[i + 3];
                l0             = sq_acc<FMA>(d0, l0);
                l1             = sq_acc<FMA>(d1, l1);
                l2             = sq_acc<FMA>(d2, l2);
                l3             = sq_acc<FMA>(d3, l3);
            }
            float dist = 0.f;
            dist       = dist + ((l0 + l1) + (l2 + l3));
            for (int i = eff; i < dim; ++i) {
                const float df = (mu[i] - x[i]) * is[i];
                dist           = sq_acc<FMA>(df, dist);
            }
            st.add(k_c64[k], 0.f, dist, k - k0);
        }
    }
    best[t] = st.idx;
    if (score)
        score[t] = st.result();
}

// u32 best densities -> bytes (paths without a byte-writing kernel; 0xffffffff -> 0xff)
__global__ __launch_bounds__(256) void best_narrow_kernel(const uint32_t* __restrict__ in, unsigned char* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (unsigned char)min(in[i], 255u);
}

// ---- Viterbi training statistics (Mm/AbstractMixtureSetEstimator.cc:117-125): one wavefront per frame, lane = dim.
// The density chosen for frame t is best_density[t][mixture[t]] (what amx_gmm_score_dev wrote).  Sums are f64 like
// Mm::Sum; with a pooled covariance every frame hits the same row, so each wave first sums its frames in
// registers and issues one atomic per dimension at the end.
// best8 (nullable): the byte form of the best-density matrix (amx_gmm_score_stats_u8_dev; 0xff widens to the u32 form's 0xffffffff)
__global__ __launch_bounds__(256) void gmm_accumulate_kernel(const float* __restrict__ feats, const uint32_t* __restrict__ mixture,
                                                            const uint32_t* __restrict__ best, const unsigned char* __restrict__ best8,
                                                            int best_ld, int T, int dim, int n_mix,
                                                            const uint32_t* __restrict__ mix_off, const uint32_t* __restrict__ k_dens,
                                                            const uint32_t* __restrict__ d_mean, const uint32_t* __restrict__ d_cov,
                                                            double* __restrict__ acc, long long off_mw, long long off_ms,
                                                            long long off_cw, long long off_cs, int pooled) {
    // A workgroup owns 256 consecutive frames.  Frames aligned to the same (state, density) are chained first and summed in
    // registers, so a density that wins many frames of the block costs ONE set of atomics (aligned speech is bursty: the
    // per-frame version spent its time on serialised f64 atomics to a handful of hot rows).
    __shared__ uint32_t s_k[256], s_mi[256], s_ci[256];
    __shared__ short    s_next[256];
    __shared__ unsigned char s_lead[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * 256;
    {
        const int t = t0 + tid;
        uint32_t  k = 0xffffffffu, mi = 0, ci = 0;
        // A frame without a density is skipped (k stays 0xffffffff: neither a lead nor chained): a mixture index outside the model,
        // or "no density" -- 0xffffffff / 0xff is what amx_gmm_score_dev and amx_gmm_best_density_dev write for a frame no density
        // ever beat FLT_MAX on (NaN / over-range features) -- or any index behind the mixture's last density
        if (t < T && mixture[t] < (uint32_t)n_mix) {
            const uint32_t m  = mixture[t];
            uint32_t       kk;
            if (best8) {
                const unsigned char b = best_ld > 0 ? best8[(size_t)t * best_ld + m] : best8[t];
                kk                    = b == 0xffu ? 0xffffffffu : (uint32_t)b;
            }
            else
                kk = best_ld > 0 ? best[(size_t)t * best_ld + m] : best[t];
            if (kk < mix_off[m + 1] - mix_off[m]) {
                k                = mix_off[m] + kk;
                const uint32_t d = k_dens[k];
                mi               = d_mean[d];
                ci               = d_cov[d];
            }
        }
        s_k[tid]  = k;
        s_mi[tid] = mi;
        s_ci[tid] = ci;
    }
    __syncthreads();
    {
        const uint32_t k = s_k[tid];
        bool lead = k != 0xffffffffu;
        for (int j = 0; j < tid && lead; ++j)
            lead = s_k[j] != k;
        int next = -1;
        if (k != 0xffffffffu)
            for (int j = tid + 1; j < 256; ++j)
                if (s_k[j] == k) {
                    next = j;
                    break;
                }
        s_lead[tid] = lead ? 1 : 0;
        s_next[tid] = (short)next;
    }
    __syncthreads();
    double pc[4] = {0, 0, 0, 0};  // pooled covariance: private partial sums for up to 4 x 64 dims
    double pcw   = 0;
    for (int j = wave; j < 256; j += 4) {
        if (!s_lead[j])
            continue;
        const uint32_t k = s_k[j], mi = s_mi[j], ci = s_ci[j];
        double         sx[4] = {0, 0, 0, 0}, sxx[4] = {0, 0, 0, 0};
        int            count = 0;
        for (int m = j; m >= 0; m = s_next[m]) {
            ++count;
            for (int i = lane, c = 0; i < dim && c < 4; i += 64, ++c) {
                const double y = (double)feats[(size_t)(t0 + m) * dim + i];
                sx[c] += y;
                sxx[c] += y * y;
            }
            for (int i = lane + 256; i < dim; i += 64) {  // dimensions beyond 256: straight atomics
                const double y = (double)feats[(size_t)(t0 + m) * dim + i];
                atomicAdd(&acc[off_ms + (long long)mi * dim + i], y);
                atomicAdd(&acc[off_cs + (pooled ? 0 : (long long)ci * dim) + i], y * y);
            }
        }
        if (lane == 0) {
            atomicAdd(&acc[k], (double)count);
            atomicAdd(&acc[off_mw + mi], (double)count);
            if (!pooled)
                atomicAdd(&acc[off_cw + ci], (double)count);
        }
        pcw += (double)count;
        for (int i = lane, c = 0; i < dim && c < 4; i += 64, ++c) {
            atomicAdd(&acc[off_ms + (long long)mi * dim + i], sx[c]);
            if (pooled)
                pc[c] += sxx[c];
            else
                atomicAdd(&acc[off_cs + (long long)ci * dim + i], sxx[c]);
        }
    }
    if (pooled) {
        for (int i = lane, c = 0; i < dim && c < 4; i += 64, ++c)
            if (pc[c] != 0.0)
                atomicAdd(&acc[off_cs + i], pc[c]);
        if (lane == 0 && pcw != 0.0)
            atomicAdd(&acc[off_cw], pcw);
    }
}

// ---- weighted Viterbi / Baum-Welch statistics (Mm/AbstractMixtureSetEstimator.cc:127-147): one wavefront per frame.
// Baum-Welch: lane = density computes s_k = 0.5 * ((m2lw + logNorm) + distance) in f32 with the reference's distance order
// (GaussDiagonalSumFeatureScorer::calculateScoresAndNumberOfDensities, Mm/GaussDiagonalMaximumFeatureScorer.cc:239-262), the
// wave takes the minimum, sums exp(best - s_k) in density order like calculateScoreAndDensity (:264-289), and every density
// whose weight * exp(score - s_k) exceeds f32 epsilon adds weight * x and (weight * x) * x to its rows (lane = dimension).
// Pooled covariance: all frames hit one row, so a wave keeps that row's sums in registers over its frames.
constexpr int kBwMaxDens   = 4096;  // densities per mixture the LDS score buffer holds (4 waves x 16 KB)
constexpr int kBwFramesPerWave = 16;

template<bool FMA>
__global__ __launch_bounds__(256) void gmm_accumulate_weighted_kernel(
        int mode, const float* __restrict__ feats, const uint32_t* __restrict__ mixture, const double* __restrict__ weight,
        const uint32_t* __restrict__ best, int best_ld, int T, int dim, int n_mix, const uint32_t* __restrict__ mix_off,
        const uint32_t* __restrict__ k_dens, const uint32_t* __restrict__ d_mean, const uint32_t* __restrict__ d_cov,
        const float* __restrict__ k_c32, const float* __restrict__ means, const float* __restrict__ isr, double* __restrict__ acc,
        long long off_mw, long long off_ms, long long off_cw, long long off_cs, int pooled) {
    __shared__ float s_p[4][kBwMaxDens];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float*    sp   = s_p[wave];
    double    pc[4] = {0, 0, 0, 0};
    double    pcw   = 0;
    const int t_begin = (blockIdx.x * 4 + wave) * kBwFramesPerWave;
    for (int t = t_begin; t < min(t_begin + kBwFramesPerWave, T); ++t) {
        const uint32_t m  = mixture[t];
        if (m >= (uint32_t)n_mix)  // wave-uniform: a frame aligned to no mixture of this model contributes nothing
            continue;
        const uint32_t k0 = mix_off[m], nd = mix_off[m + 1] - k0;
        const float*   x  = feats + (size_t)t * dim;
        const double   w  = weight ? weight[t] : 1.0;
        uint32_t       j_lo = 0, j_hi = nd;
        if (mode == 1) {
            float lmin = FLT_MAX;
            for (uint32_t j = lane; j < nd; j += 64) {
                const uint32_t d  = k_dens[k0 + j];
                const float*   mu = means + (size_t)d_mean[d] * dim;
                const float*   is = isr + (size_t)d_cov[d] * dim;
                float          l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
                const int      eff = dim & ~3;
                for (int i = 0; i < eff; i += 4) {
                    const float d0 = (mu[i] - x[i]) * is[i], d1 = (mu[i + 1] - x[i + 1]) * is[i + 1];
                    const float d2 = (mu[i + 2] - x[i + 2]) * is[i + 2], d3 = (mu[i + 3] - x[i + 3]) * is[i + 3];
                    l0 = sq_acc<FMA>(d0, l0);
                    l1 = sq_acc<FMA>(d1, l1);
                    l2 = sq_acc<FMA>(d2, l2);
                    l3 = sq_acc<FMA>(d3, l3);
                }
                float dist = 0.f;
                dist       = dist + ((l0 + l1) + (l2 + l3));
                for (int i = eff; i < dim; ++i) {
                    const float df = (mu[i] - x[i]) * is[i];
                    dist           = sq_acc<FMA>(df, dist);
                }
                const float sk = 0.5f * (k_c32[k0 + j] + dist);
                sp[j]          = sk;
                lmin           = fminf(lmin, sk);
            }
            for (int o = 32; o > 0; o >>= 1)
                lmin = fminf(lmin, __shfl_xor(lmin, o));
            __builtin_amdgcn_wave_barrier();  // LDS traffic of one wave is ordered; this only pins the compiler's schedule
            float sum_exp = 0.f;
            for (uint32_t j = 0; j < nd; ++j)  // density order, like the reference's loop (all lanes compute the same sum)
                sum_exp += expf(lmin - sp[j]);
            const float log_den = lmin - logf(sum_exp);
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = lane; j < nd; j += 64)
                sp[j] = expf(log_den - sp[j]);  // the posterior; each lane rewrites only its own entries
        }
        else {
            const uint32_t kk = best_ld > 0 ? best[(size_t)t * best_ld + m] : best[t];
            if (kk >= nd)  // "no density" (0xffffffff: a NaN / over-range frame) or an index behind the mixture's last density
                continue;
            j_lo              = kk;
            j_hi              = kk + 1;
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j = j_lo; j < j_hi; ++j) {
            double fw = w;
            if (mode == 1) {
                fw = w * (double)sp[j];
                if (!(fw > (double)FLT_EPSILON))
                    continue;
            }
            const uint32_t k  = k0 + j;
            const uint32_t d  = k_dens[k];
            const uint32_t mi = d_mean[d], ci = d_cov[d];
            if (lane == 0) {
                atomicAdd(&acc[k], fw);
                atomicAdd(&acc[off_mw + mi], fw);
                if (!pooled)
                    atomicAdd(&acc[off_cw + ci], fw);
            }
            pcw += fw;
            for (int i = lane, c = 0; i < dim; i += 64, ++c) {
                const double y  = (double)x[i];
                const double wy = fw * y;
                atomicAdd(&acc[off_ms + (long long)mi * dim + i], wy);
                if (pooled && c < 4)
                    pc[c] += wy * y;
                else
                    atomicAdd(&acc[off_cs + (long long)ci * dim + i], wy * y);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (pooled) {
        for (int i = lane, c = 0; i < dim && c < 4; i += 64, ++c)
            if (pc[c] != 0.0)
                atomicAdd(&acc[off_cs + i], pc[c]);
        if (lane == 0 && pcw != 0.0)
            atomicAdd(&acc[off_cw], pcw);
    }
}

// ---- two-stage path for tied models
struct GmmDistParams {
    const float* __restrict__ feats;     // [T x dim] (chunk)
    float* __restrict__ dist;            // [n_dens x Tpad]
    const uint32_t* __restrict__ d_mean; // [n_dens]
    const uint32_t* __restrict__ d_cov;  // [n_dens]
    const float* __restrict__ means;
    const float* __restrict__ isr;
    int T, Tpad, dim, n_dens, dens_tile;
};

struct GmmDistDims {
    int T, Tpad, dim, n_dens, dens_tile;
};

// STAGE: the workgroup's 256 frames come in as ONE coalesced run through LDS.  Lane = frame reads feats[t][i] with a stride of dim
// floats -- 64 separate 64-byte sectors per load instruction, ~150 cycles of the CU's address unit each (tools/gather_probe.hip) --
// which is noise behind a long density loop but most of the kernel when a small batch is cut into many short workgroups.
template<int DIM, bool STAGE = false, bool FMA = false>
__global__ __launch_bounds__(256) void gmm_dist_kernel(const float* __restrict__ g_feats, float* __restrict__ g_dist, double* __restrict__ g_dist64,
                                                      const uint32_t* __restrict__ g_d_mean, const uint32_t* __restrict__ g_d_cov,
                                                      const float* __restrict__ g_means, const float* __restrict__ g_isr,
                                                      GmmDistDims dims, float* __restrict__ g_dt = nullptr,
                                                      const uint32_t* __restrict__ g_pos = nullptr, int dt_ld = 0) {
    struct {
        const float* __restrict__ feats; float* __restrict__ dist; const uint32_t* __restrict__ d_mean;
        const uint32_t* __restrict__ d_cov; const float* __restrict__ means; const float* __restrict__ isr;
        int T, Tpad, dim, n_dens, dens_tile;
    } p = {g_feats, g_dist, g_d_mean, g_d_cov, g_means, g_isr, dims.T, dims.Tpad, dims.dim, dims.n_dens, dims.dens_tile};
    extern __shared__ float xs_all[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int t    = (blockIdx.y * 4 + wave) * 64 + lane;
    const int tt   = t < p.T ? t : (p.T - 1);
    float     x[DIM > 0 ? DIM : 1];
    float*    xs = nullptr;
    if (DIM > 0 && STAGE) {
        constexpr int LD   = (DIM > 0 ? DIM : 1) + 1;  // odd row stride: the lanes' reads spread over the banks
        const int     base = blockIdx.y * 256;
        for (int e = threadIdx.x; e < 256 * DIM; e += 256) {
            const int r = e / DIM, i = e - r * DIM;
            xs_all[r * LD + i] = p.feats[(size_t)min(base + r, p.T - 1) * DIM + i];  // past the end: the last frame, as below
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < DIM; ++i)
            x[i] = xs_all[(wave * 64 + lane) * LD + i];
    }
    else if (DIM > 0) {
#pragma unroll
        for (int i = 0; i < DIM; ++i)
            x[i] = p.feats[(size_t)tt * DIM + i];
    }
    else {
        xs = xs_all + wave * 64 * p.dim + lane;
        for (int i = 0; i < p.dim; ++i)
            xs[i * 64] = p.feats[(size_t)tt * p.dim + i];
    }
    const int d0 = blockIdx.x * p.dens_tile;
    const int d1 = min(d0 + p.dens_tile, p.n_dens);
    for (int db = d0; db < d1; db += 4) {  // four densities per trip: their frame-major pieces leave as one 16-byte store
        float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = db + j;
            if (d >= d1)
                break;
            const float* mu = p.means + (size_t)p.d_mean[d] * (DIM > 0 ? DIM : p.dim);
            const float* is = p.isr + (size_t)p.d_cov[d] * (DIM > 0 ? DIM : p.dim);
            float        dist;
            if (DIM > 0)
                dist = gmm_distance<(DIM > 0 ? DIM : 1), FMA>(x, mu, is);
            else
                dist = gmm_distance_rt<FMA>(xs, p.dim, mu, is);
            q[j] = dist;
            if (t < p.Tpad) {
                p.dist[(size_t)d * p.Tpad + t] = dist;  // coalesced along t
                if (g_dist64)
                    g_dist64[(size_t)d * p.Tpad + t] = (double)dist;
            }
        }
        // shared-list models, pruned scorer (gmm_tied.hip): the frame-major image dt[t][list position] it works on, written here
        // instead of by a transposing kernel (10 us, all of it the latency of two dependent trips).  A lane's piece is a sector of its
        // own (64 sectors per store instruction, ~150 cycles of the CU's address unit), hence four list-adjacent densities at once.
        if (g_dt && t < p.T) {
            const uint32_t p0 = g_pos[db];  // wave-uniform
            const bool     quad = db + 3 < d1 && p0 != 0xffffffffu && (p0 & 3u) == 0u && g_pos[db + 1] == p0 + 1u && g_pos[db + 2] == p0 + 2u &&
                              g_pos[db + 3] == p0 + 3u;
            if (quad)
                *(float4*)(g_dt + (size_t)t * dt_ld + p0) = make_float4(q[0], q[1], q[2], q[3]);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (db + j < d1) {
                        const uint32_t pos = g_pos[db + j];
                        if (pos != 0xffffffffu)
                            g_dt[(size_t)t * dt_ld + pos] = q[j];
                    }
            }
        }
    }
}

// The pruned scorer of a shared-list model (gmm_tied.hip) only reads the frame-major image dt[t][list position], and a decoder-sized
// batch is ONE frame block of gmm_dist_kernel: 1024 workgroups that each stage 40 KB of frames for four densities, 20 us for 3 us of
// arithmetic.  Here the roles are swapped -- lane = list position, the model's means / inverse deviations come from tables transposed
// at creation ([dim][Kpad], coalesced, 2 x DIM registers per lane for the wave's whole life), the frame is wave-uniform and arrives
// through the scalar cache, and the row dt[t][k0 .. k0 + 63] leaves as one 256-byte store.  Same operation order per (density, frame)
// as gmm_distance: bit-identical distances.  Nothing density-major is written (no other kernel of the pruned path reads it).
// NEAR: the kernel also keeps, per (frame, lane = list position mod 64), the smallest key (distance bits << 32) | position -- the
// frame's closest density of every residue class, which the pruned scorer's bounds start from (tied_near_kernel, gmm_tied.hip, then
// does not run: a launch and a pass over the image less).  The four waves' keys of a frame meet in LDS, one wave takes the minimum to
// the frame's 64 keys with atomics; the keys are in their empty state (+inf, 0) when the kernel starts (tied_list_kernel puts them back).
template<int DIM, bool FMA, bool NEAR>
__global__ __launch_bounds__(256) void gmm_dist_list_kernel(const float* __restrict__ g_feats, const float* __restrict__ g_means_t,
                                                           const float* __restrict__ g_isr_t, int K, int Kpad, int T, int frames,
                                                           float* __restrict__ g_dt, unsigned long long* __restrict__ g_near) {
    constexpr unsigned long long kEmpty = 0x7f80000000000000ull;
    __shared__ unsigned long long s_key[NEAR ? 2 * 256 : 1];
    const int k  = blockIdx.x * 256 + threadIdx.x;
    const int kk = k < Kpad ? k : Kpad - 1;
    float     mu[DIM], is[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
        mu[i] = g_means_t[(size_t)i * Kpad + kk];
        is[i] = g_isr_t[(size_t)i * Kpad + kk];
    }
    const int t0 = blockIdx.y * frames, t1 = min(t0 + frames, T);
    for (int t = t0; t < t1; ++t) {
        const float* __restrict__ xr = g_feats + (size_t)t * DIM;  // wave-uniform: scalar loads
        float x[DIM];
#pragma unroll
        for (int i = 0; i < DIM; ++i)
            x[i] = xr[i];
        const float dist = gmm_distance<DIM, FMA>(x, mu, is);
        if (k < K)
            g_dt[(size_t)t * Kpad + k] = dist;
        if (NEAR) {
            const int par = (t - t0) & 1;  // two buffers: the wave that reduces frame t may still read while frame t + 1 is written
            s_key[par * 256 + threadIdx.x] = k < K ? ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned)k : kEmpty;
            __syncthreads();
            if ((int)(threadIdx.x >> 6) == ((t - t0) & 3)) {
                const unsigned long long* p = s_key + par * 256 + (threadIdx.x & 63);
                const unsigned long long  a = p[0] < p[64] ? p[0] : p[64], b = p[128] < p[192] ? p[128] : p[192];
                const unsigned long long  m = a < b ? a : b;
                if (m < kEmpty)
                    __hip_atomic_fetch_min(g_near + (size_t)t * 64 + (threadIdx.x & 63), m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

struct GmmCombineParams {
    const float* __restrict__ dist;        // [n_dens x Tpad]
    float* __restrict__ scores;            // [T x n_mix]
    uint32_t* __restrict__ best;           // nullable
    const uint32_t* __restrict__ mix_off;
    const uint32_t* __restrict__ k_dens;   // [nk] density of entry k
    const double* __restrict__ k_c64;
    const float* __restrict__ k_c32;
    int T, Tpad, n_mix, mix_tile;
};

// ---- uniform-list tied models: every mixture weights the SAME density list (tied-mixture / semi-continuous
// systems).  lane = mixture, FR frames per pass live in registers; the distances of the current density are
// wave-uniform and arrive through the scalar cache, the weights are read once per pass, transposed [k][mixture]
// (coalesced).  Scores are written coalesced along the mixture index.
struct GmmUniformDims {
    int T, Tpad, n_mix, mix_pad, K, t0;
};

template<class State, int FR>
__global__ __launch_bounds__(256) void gmm_combine_uniform_kernel(const float* __restrict__ g_dist, const double* __restrict__ g_dist64,
                                                                 float* __restrict__ g_scores, uint32_t* __restrict__ g_best,
                                                                 const float* __restrict__ g_m2lw_t, const uint32_t* __restrict__ g_k_dens,
                                                                 const double* __restrict__ g_ln64, const float* __restrict__ g_ln32,
                                                                 GmmUniformDims dims) {
    const int  m    = blockIdx.x * 256 + threadIdx.x;
    const int  t0   = blockIdx.y * FR;  // within the chunk
    const bool live = m < dims.n_mix;
    const int  mm   = live ? m : dims.n_mix - 1;
    State      st[FR];
    for (int k = 0; k < dims.K; ++k) {
        const float    w   = g_m2lw_t[(size_t)k * dims.mix_pad + mm];
        const uint32_t d   = g_k_dens[k];  // wave-uniform
        const size_t   row = (size_t)d * dims.Tpad + t0;
        if (State::kF64) {
            const double c64 = (double)w + g_ln64[k];
#pragma unroll
            for (int f = 0; f < FR; ++f)
                st[f].add_d(c64, g_dist64[row + f], (uint32_t)k);
        }
        else {
            const float c32 = w + g_ln32[k];
#pragma unroll
            for (int f = 0; f < FR; ++f)
                st[f].add(0.0, c32, g_dist[row + f], (uint32_t)k);
        }
    }
    if (live) {
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            const int t = t0 + f;
            if (t < dims.T) {
                g_scores[(size_t)t * dims.n_mix + m] = State::kF64 ? st[f].result_d() : st[f].result();
                if (g_best)
                    g_best[(size_t)t * dims.n_mix + m] = st[f].idx;
            }
        }
    }
}

// ---- screened maximum approximation for uniform-list tied models.
// The reference's per-(frame, mixture) loop is `s = (m2lw + logNorm) + dist` in f64 and `if ((double)best > s) { best =
// (float)s; idx = k; }` (Mm/GaussDiagonalMaximumFeatureScorer.cc:116-141).  Because f32 rounding is monotone, the final
// best is f32(min_k s_k) = b, and the final index is the last k that is either the first k with f32(s_k) == b or has
// s_k < (double)b -- both sets contain only densities whose sum rounds to b.  So running the very same sequential rule
// over ANY subsequence that contains every k with f32(s_k) == b gives bit-identical (best, idx).
// Pass 1 finds an f32 approximation of the minimum (1 add + 1 min per density and frame), pass 2 recomputes the f32
// sum and applies the exact f64 rule only to densities within tau of it (1 add + 1 compare; the exact branch runs for
// ~1 density per frame and mixture).  |s^ - s| <= 2^-24 (|a^| + |s^|) for a^ = fl32(m2lw + logNorm), s^ = fl32(a^ + dist),
// and two sums that round to the same f32 differ by <= ulp32(b), hence tau = 2^-21 (max_k |a^| + |min s^|) covers every
// such k with a 2x margin.  Replaces 4 f64-rate + 3 f32 operations per density by 4 f32 operations.
typedef float gmm_f32x2 __attribute__((ext_vector_type(2)));

// ---- register-tiled (min,+) product carrying that screen: C[t][m] = min_k (a^[k][m] + dist[k][t]), a^ tabulated
// [K][mix_pad] next to the weights, max_k |a^| per mixture tabulated too.
// Neither operand may come through the scalar cache (measured: a lane = frame variant with the a^ rows as SGPR operands
// was bound by scalar-cache misses at ~0.25 TB/s chip-wide, 2.2 ms; a lane = mixture variant re-streamed the 164 MB
// table once per 8 frames, 2.7 ms; the plain f64 kernel below 2.53 ms; this kernel 1.7 ms).  Both operands are staged in
// LDS like GEMM operands: a workgroup owns 64 mixtures x 64 frames, a thread 4 x 4 of them, and a
// K-chunk of 64 densities (16 KB of a^ rows + 16 KB of distance rows, coalesced 256-byte rows) is fetched into registers
// while the previous chunk is consumed from LDS.  Per density a lane reads one float4 of each operand (16 / 4 distinct
// addresses per wave, conflict free) for 16 sums: 8 packed adds + 8 v_min3 in pass 1; pass 2 forms min(s^ - thr) the
// same way and enters the exact f64 rule only where it is <= 0.
constexpr int TT_KB = 64;  // densities per chunk

__global__ __launch_bounds__(256) void gmm_tied_tile_kernel(const float* __restrict__ g_dist, float* __restrict__ g_scores,
                                                           uint32_t* __restrict__ g_best, const float* __restrict__ g_m2lw_t,
                                                           const float* __restrict__ g_ahat_t, const float* __restrict__ g_amax,
                                                           const uint32_t* __restrict__ g_k_dens, const double* __restrict__ g_ln64,
                                                           GmmUniformDims dims) {
    __shared__ __attribute__((aligned(16))) float sA[TT_KB][64];
    __shared__ __attribute__((aligned(16))) float sD[TT_KB][64];
    const int tid = threadIdx.x;
    const int mi = tid & 15, ti = tid >> 4;  // thread tile: mixtures 4*mi.., frames 4*ti..
    const int m_blk = blockIdx.x * 64, t_blk = blockIdx.y * 64;
    const unsigned Tp = (unsigned)dims.Tpad, mp = (unsigned)dims.mix_pad;
    const int n_chunks = (dims.K + TT_KB - 1) / TT_KB;
    // staging: 64 rows x 256 B per operand = 1024 float4; thread -> rows r0 + 16*i (i < 4), float4 column c4
    const int r0 = tid >> 4, c4 = (tid & 15) * 4;
    float4    ra[4], rd[4];
    auto fetch = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = chunk * TT_KB + r0 + 16 * i;
            if (k < dims.K) {
                ra[i] = *(const float4*)(g_ahat_t + (size_t)k * mp + m_blk + c4);
                rd[i] = *(const float4*)(g_dist + g_k_dens[k] * Tp + t_blk + c4);
            }
            else {  // beyond the density list: sums that can never win or pass the threshold
                ra[i] = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX);
                rd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(float4*)&sA[r0 + 16 * i][c4] = ra[i];
            *(float4*)&sD[r0 + 16 * i][c4] = rd[i];
        }
    };

    // ---------------- pass 1: f32 minimum
    float mn[4][4];  // [mixture][frame]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            mn[i][j] = FLT_MAX;
    fetch(0);
    for (int c = 0; c < n_chunks; ++c) {
        __syncthreads();  // everybody is done with the previous chunk
        stash();
        __syncthreads();
        if (c + 1 < n_chunks)
            fetch(c + 1);
        // two densities per v_min3; the fragments of the next pair are read while this pair is folded
        float4 xa0, xa1, xd0, xd1, ya0, ya1, yd0, yd1;
#define TT_LOAD(P, K)                                  \
    P##a0 = *(const float4*)&sA[(K)][4 * mi];          \
    P##a1 = *(const float4*)&sA[(K) + 1][4 * mi];      \
    P##d0 = *(const float4*)&sD[(K)][4 * ti];          \
    P##d1 = *(const float4*)&sD[(K) + 1][4 * ti];
#define TT_FOLD(P)                                                                                                          \
    {                                                                                                                       \
        const float av0[4] = {P##a0.x, P##a0.y, P##a0.z, P##a0.w}, av1[4] = {P##a1.x, P##a1.y, P##a1.z, P##a1.w};           \
        const float dv0[4] = {P##d0.x, P##d0.y, P##d0.z, P##d0.w}, dv1[4] = {P##d1.x, P##d1.y, P##d1.z, P##d1.w};           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; j += 2) {                    \
            const gmm_f32x2 s0 = gmm_f32x2{dv0[j], dv0[j + 1]} + gmm_f32x2{av0[i], av0[i]};                                 \
            const gmm_f32x2 s1 = gmm_f32x2{dv1[j], dv1[j + 1]} + gmm_f32x2{av1[i], av1[i]};                                 \
            mn[i][j]           = min3_raw(mn[i][j], s0.x, s1.x);                                                            \
            mn[i][j + 1]       = min3_raw(mn[i][j + 1], s0.y, s1.y);                                                        \
        }                                                                                                                   \
    }
        TT_LOAD(x, 0)
#pragma unroll 2
        for (int k = 0; k < TT_KB; k += 4) {
            TT_LOAD(y, k + 2)
            TT_FOLD(x)
            if (k + 4 < TT_KB) {
                TT_LOAD(x, k + 4)
            }
            TT_FOLD(y)
        }
#undef TT_LOAD
#undef TT_FOLD
    }
    float thr[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float am = g_amax[m_blk + 4 * mi + i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            thr[i][j] = mn[i][j] + (4.76837158e-7f * (am + fabsf(mn[i][j])) + 1e-30f);  // tau = 2^-21 (...)
    }

    // ---------------- pass 2: exact rule on the densities within tau
    MaxState st[4][4];
    fetch(0);
    for (int c = 0; c < n_chunks; ++c) {
        __syncthreads();
        stash();
        __syncthreads();
        if (c + 1 < n_chunks)
            fetch(c + 1);
#pragma unroll 4
        for (int k = 0; k < TT_KB; ++k) {
            const float4 a4 = *(const float4*)&sA[k][4 * mi];
            const float4 d4 = *(const float4*)&sD[k][4 * ti];
            const float  av[4] = {a4.x, a4.y, a4.z, a4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
            float        worst = FLT_MAX;  // min over the tile of s^ - thr  (x - y <= 0 <=> x <= y exactly)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const gmm_f32x2 e = (gmm_f32x2{dv[j], dv[j + 1]} + gmm_f32x2{av[i], av[i]}) - gmm_f32x2{thr[i][j], thr[i][j + 1]};
                    worst             = min3_raw(worst, e.x, e.y);
                }
            const int kk = c * TT_KB + k;
            // kk < K: the padding rows (a^ = FLT_MAX) can only "pass" when a frame's sums are all inf / NaN (a non-finite feature);
            // the reference then keeps its initial (FLT_MAX, no density), and so must this
            if (worst <= 0.f && kk < dims.K) {
                const double ln = g_ln64[kk];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bool hit = false;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        hit |= (dv[j] + av[i]) <= thr[i][j];
                    if (hit) {
                        const double c64 = (double)g_m2lw_t[(size_t)kk * mp + m_blk + 4 * mi + i] + ln;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if ((dv[j] + av[i]) <= thr[i][j])
                                st[i][j].add(c64, 0.f, dv[j], (uint32_t)kk);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = t_blk + 4 * ti + j;
        if (t >= dims.T)
            continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_blk + 4 * mi + i;
            if (m < dims.n_mix) {
                g_scores[(size_t)t * dims.n_mix + m] = st[i][j].result();
                if (g_best)
                    g_best[(size_t)t * dims.n_mix + m] = st[i][j].idx;
            }
        }
    }
}

// =================================================================================================================
// MFMA-screened maximum approximation for mixtures with PRIVATE densities (CART-style models, <= 16 densities each).
//
// gmm_direct_kernel evaluates every density exactly: 4 unfused f32 operations per (frame, density, dimension) in the
// reference's association order, which already runs near the unpacked VALU issue ceiling.  But only the densities whose
// f64 sum rounds to the winning f32 value can influence (score, best density) -- see the subsequence argument in front
// of gmm_tied_tile_kernel -- and those can be found with arithmetic that need not be exact, as long as its error is
// bounded.  Expanding ((mu - x) r)^2 turns the distance into a dot product plus a per-density constant (the frame term is
// common to a mixture's densities and drops out of the comparison):
//     pooled covariance      g[t][k] = c_k + sum_i (-2 mu_ki r_i)   * (x_ti r_i)                   K = dim
//     per-density covariance g[t][k] = c_k + sum_i (-2 mu_ki r_ki^2) * x_ti + (r_ki^2) * x_ti^2    K = 2 dim
// with c_k = m2lw_k + logNorm_k + sum_i (mu_ki r_ki)^2.  That is a GEMM with f16 operands on the matrix cores
// (gmm_screen_kernel, 256 density slots x 256 frames per workgroup, mixtures padded to 16 slots).  Its epilogue takes the
// minimum over each mixture's 16 slots and emits a 16-bit candidate mask: slot j is kept unless g_j > g_min + tau, where
//     tau = 2.05 (ra nx + NA rx) + 1.3e-4 sqrt(K) (na + nx) + 1.6e-5 (|g_min| + max|c| + q)
// bounds twice the worst difference between g and the reference's own f64 sum minus the frame term.  f16 rounding of the operands:
// with A = A^ + dA, X = X^ + dX (the rounded rows and their residuals) the dot product is off by exactly dA.X^ + A.dX, at most
// ||dA|| ||X^|| + ||A|| ||dX|| (Cauchy-Schwarz) -- ra = the largest residual norm among the mixture's rows (host, f64), nx = ||X^||,
// rx = ||dX|| (pack kernel: v - (f32)(f16)v is exact), NA = the largest row norm of the model; round 1 used the worst case
// 2^-11 ||A|| for both residuals, which is 2.4 times the typical one and kept 2.3 times as many further survivors.  2^-14 absolute
// per element if subnormals were flushed, f32 accumulation of <= 128 exact products, rounding of c_k, the (dim + 3) ulp error of
// the reference's f32 distance, plus one ulp of the winning f32 (q bounds the dropped frame term).  Frames whose operand
// does not fit f16 get nx = inf and therefore all-ones masks; models whose operand does not fit are not screened at all.
// gmm_screen_exact_kernel then runs gmm_distance and the reference's sequential f64 rule over the surviving slots only
// (~1 of 16), in slot order: bit-identical scores and density indices at ~1/10 of the exact arithmetic.
typedef _Float16 gmm_f16x8 __attribute__((ext_vector_type(8)));
typedef float    gmm_f32x16 __attribute__((ext_vector_type(16)));

struct GmmScreenDims {
    int   T, Tpad, dim, Kp, Mpad16, pooled;
    float rmax2, sqrtK;
    float na_all;  // 2.05 x the largest operand-row norm of the model: weighs the frame operand's f16 residual (see the threshold)
};

// features -> f16 operand rows [Tpad x Kp] (zero padded), nx[t] = ||row|| (inf when the row does not fit f16), q[t].
// One wavefront per frame, lane = operand column (coalesced 128-byte rows; a thread-per-frame loop took 23 us at batch 256).
__global__ __launch_bounds__(256) void gmm_screen_pack_kernel(const float* __restrict__ g_feats, const float* __restrict__ g_isr0,
                                                             _Float16* __restrict__ g_X, float* __restrict__ g_nx,
                                                             float* __restrict__ g_q, GmmScreenDims d) {
    const int lane = threadIdx.x & 63;
    const int t    = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= d.Tpad)
        return;
    _Float16* row = g_X + (size_t)t * d.Kp;
    const int kd  = d.pooled ? d.dim : 2 * d.dim;  // columns kd, kd + 1 multiply the split constant c_hi, c_lo
    float     n2 = 0.f, q = 0.f, r2 = 0.f;  // r2: squared norm of the f16 rounding residual of the row (v - r is exact in f32)
    bool      fits = true;
    for (int i = lane; i < d.Kp; i += 64) {
        float v = 0.f;
        if (t < d.T) {
            if (d.pooled) {
                if (i < d.dim)
                    v = g_feats[(size_t)t * d.dim + i] * g_isr0[i];
            }
            else if (i < d.dim)
                v = g_feats[(size_t)t * d.dim + i];
            else if (i < 2 * d.dim) {
                const float x = g_feats[(size_t)t * d.dim + i - d.dim];
                v             = x * x;
            }
        }
        fits &= fabsf(v) <= 65504.f;  // false for NaN as well
        const _Float16 hv = (_Float16)v;
        const float    r  = (float)hv;
        n2 += r * r;
        r2 += (v - r) * (v - r);
        if (d.pooled || i < d.dim)
            q += v * v;
        row[i] = (i == kd || i == kd + 1) ? (_Float16)(t < d.T ? 1.f : 0.f) : hv;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n2 += __shfl_xor(n2, off, 64);
        q += __shfl_xor(q, off, 64);
        r2 += __shfl_xor(r2, off, 64);
    }
    const bool all_fit = __ballot(!fits) == 0ull;
    if (lane == 0) {
        g_nx[t] = all_fit ? sqrtf(n2) : __builtin_inff();
        g_q[t]  = 1.6e-5f * (d.pooled ? q : q * d.rmax2) + d.na_all * (sqrtf(r2) * 1.00001f);
    }
}

// epilogue of the screen: lane (tl32, hh) holds, for frame tl32 of pass j, operand rows i*32 + 8g + 4hh + e of the wave's 128:
// mixture 2i + (g >> 1) of its 8, row (g & 1) * 8 + 4hh + e within it; the partner lane (lane ^ 32) holds the other 8 rows.  The
// model side stores slot s of a mixture at row ((s >> 2) & 1) * 8 + (s >> 3) * 4 + (s & 3), which makes a lane's 8 values the
// slots 8hh .. 8hh + 7 in order (the mask byte needs no bit shuffling).
// The per-density constant already sits in the accumulators (two extra K columns c_hi + c_lo against X = 1), the threshold is
// three fused operations on per-mixture / per-frame precomputed pieces, and a survivor bit is the sign of thr - g shifted into
// the mask by v_alignbit: ~30 VALU operations per mixture and frame pair instead of ~60 (this epilogue, not the 32 MFMAs per
// wave, is what the kernel's time goes to).
__device__ __forceinline__ void gmm_screen_epilogue(const gmm_f32x16 (&acc)[4][2], const float* __restrict__ g_p1,
                                                    const float* __restrict__ g_p2, const float (&nxv)[2], const float (&qv)[2],
                                                    uint16_t* __restrict__ g_masks, int t_first, int mt0, int lane, int Mpad16) {
    const int tl32 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int   t  = t_first + j * 32 + tl32;
        const float nx = nxv[j], q = qv[j];
        const bool  all = !(nx < __builtin_inff());  // operand row did not fit f16: keep every slot
        unsigned    packed[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned two = 0;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const gmm_f32x16& c = acc[i][j];
                const int         o = gp * 8;
                float mn = min3_first(c[o], c[o + 1], c[o + 2]);
                mn       = min3_raw(mn, c[o + 3], c[o + 4]);
                mn       = min3_raw(mn, c[o + 5], c[o + 6]);
                mn       = min3_raw(mn, c[o + 7], c[o + 7]);
                mn       = min3_raw(mn, __shfl_xor(mn, 32, 64), mn);  // the partner lane's 8 slots
                const int   m   = mt0 + i * 2 + gp;
                // tau = 2.05 (ra nx + NA rx) + 1.3e-4 sqrtK (na + nx) + 1.6e-5 (|mn| + cabs + q): p1 = 2.05 ra + 1.3e-4 sqrtK,
                // p2 = 1.3e-4 sqrtK na + 1.6e-5 cabs per mixture, q (incl. 2.05 NA rx) per frame
                const float thr = mn + fmaf(nx, g_p1[m], fmaf(fabsf(mn), 1.6e-5f, g_p2[m] + q)) + 1e-30f;
                unsigned    bits = 0;  // bit k = "value k is above the threshold", filled from the top down
#pragma unroll
                for (int e = 6; e >= 0; e -= 2) {  // packed subtract, two values per instruction
                    const gmm_pk2 dd = gmm_pk2{thr, thr} - gmm_pk2{c[o + e], c[o + e + 1]};
                    bits             = __builtin_amdgcn_alignbit(bits, __float_as_uint(dd.y), 31);  // (bits << 1) | sign(thr - g)
                    bits             = __builtin_amdgcn_alignbit(bits, __float_as_uint(dd.x), 31);
                }
                // the slot rows of a mixture are stored so that this lane's 8 values are slots 8 hh .. 8 hh + 7 in order
                unsigned mine = (~bits & 0xffu) << (8 * hh);
                mine |= (unsigned)__shfl_xor((int)mine, 32, 64);
                two |= (all ? 0xffffu : mine) << (16 * gp);
            }
            packed[i] = two;
        }
        if (hh == 0)  // 8 masks x 16 bit = one 16-byte store; rows and groups are 16-byte aligned (Mpad16 % 16 == 0)
            *(uint4*)(g_masks + (size_t)t * Mpad16 + mt0) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    }
}

// LDS: [KT][A tile 256 x 128 B | X tile 256 x 128 B] [c: 256 f32]; same row swizzle as the bf16 GEMM
template<int KT>
__global__ __launch_bounds__(512) void gmm_screen_kernel(const _Float16* __restrict__ g_A, const _Float16* __restrict__ g_X,
                                                        const float* __restrict__ g_c, const float* __restrict__ g_na,
                                                        const float* __restrict__ g_cabs, const float* __restrict__ g_nx,
                                                        const float* __restrict__ g_q, uint16_t* __restrict__ g_masks, int n_tiles_r,
                                                        GmmScreenDims d) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int STAGE = 64 * 1024, A_BYTES = 32 * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wt = wave & 3;  // wave tile: 128 slots x 64 frames
    const int tile_r = blockIdx.x % n_tiles_r, tile_t = blockIdx.x / n_tiles_r;
    const int r0 = tile_r * 256, t0 = tile_t * 256;
    const int Kp = KT * 64;
    {
        const int xr = ((wave & 1) * 4 + (lane >> 4)) & 7;
        const size_t off = (size_t)(wave * 8 + (lane >> 3)) * Kp + (((lane & 7) ^ xr) << 3);
        const _Float16* pa = g_A + (size_t)r0 * Kp + off;
        const _Float16* px = g_X + (size_t)t0 * Kp + off;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            char* base = lds + kt * STAGE + wave * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_global_load_lds((const void*)(pa + (size_t)i * 64 * Kp + kt * 64),
                                                 (__attribute__((address_space(3))) void*)(base + i * 8 * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void*)(px + (size_t)i * 64 * Kp + kt * 64),
                                                 (__attribute__((address_space(3))) void*)(base + A_BYTES + i * 8 * 1024), 16, 0, 0);
            }
        }
    }
    gmm_f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frow = lane & 31, fk = lane >> 5;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const char* abase = lds + kt * STAGE;
        const char* xbase = abase + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            gmm_f16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wn * 128 + i * 32 + frow;
                a[i]        = *(const gmm_f16x8*)(abase + r * 128 + (((ks * 2 + fk) ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wt * 64 + j * 32 + frow;
                b[j]        = *(const gmm_f16x8*)(xbase + r * 128 + (((ks * 2 + fk) ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    const float nxv[2] = {g_nx[t0 + wt * 64 + (lane & 31)], g_nx[t0 + wt * 64 + 32 + (lane & 31)]};
    const float qv[2]  = {g_q[t0 + wt * 64 + (lane & 31)], g_q[t0 + wt * 64 + 32 + (lane & 31)]};
    gmm_screen_epilogue(acc, g_na, g_cabs, nxv, qv, g_masks, t0 + wt * 64, tile_r * 16 + wn * 8, lane, d.Mpad16);
}

// K <= 64 (pooled covariance, dim <= 64): the frame tile stays in LDS and the workgroup walks a range of slot tiles, the next
// tile's operand (32 KB) arriving while the current one is multiplied and reduced.  [X 32 KB][A x2 32 KB][c x2 1 KB]
__global__ __launch_bounds__(512) void gmm_screen_persist_kernel(const _Float16* __restrict__ g_A, const _Float16* __restrict__ g_X,
                                                                const float* __restrict__ g_c, const float* __restrict__ g_na,
                                                                const float* __restrict__ g_cabs, const float* __restrict__ g_nx,
                                                                const float* __restrict__ g_q, uint16_t* __restrict__ g_masks, int n_tiles_r,
                                                                int r_split, GmmScreenDims d) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int TILE = 32 * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wt = wave & 3;
    const int tile_t = blockIdx.x / r_split, part = blockIdx.x % r_split;
    const int per = (n_tiles_r + r_split - 1) / r_split;
    const int r_begin = part * per, r_end = min(n_tiles_r, r_begin + per);
    if (r_begin >= r_end)
        return;
    const int t0 = tile_t * 256;
    const int xr = ((wave & 1) * 4 + (lane >> 4)) & 7;
    const size_t off = (size_t)(wave * 8 + (lane >> 3)) * 64 + (((lane & 7) ^ xr) << 3);
    auto load_tile = [&](const _Float16* base, char* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(base + off + (size_t)i * 64 * 64),
                                             (__attribute__((address_space(3))) void*)(dst + wave * 1024 + i * 8 * 1024), 16, 0, 0);
    };
    load_tile(g_X + (size_t)t0 * 64, lds);
    load_tile(g_A + (size_t)r_begin * 256 * 64, lds + TILE);
    const float nxv[2] = {g_nx[t0 + wt * 64 + (lane & 31)], g_nx[t0 + wt * 64 + 32 + (lane & 31)]};
    const float qv[2]  = {g_q[t0 + wt * 64 + (lane & 31)], g_q[t0 + wt * 64 + 32 + (lane & 31)]};
    const int   frow = lane & 31, fk = lane >> 5;
    for (int r = r_begin; r < r_end; ++r) {
        const int buf = (r - r_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // my share of tile r is in LDS
        __builtin_amdgcn_s_barrier();                                // ... everybody's is, and buffer buf^1 is free again
        if (r + 1 < r_end) {
            load_tile(g_A + (size_t)(r + 1) * 256 * 64, lds + TILE + (buf ^ 1) * TILE);
        }
        gmm_f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    acc[i][j][q] = 0.f;
        const char* abase = lds + TILE + buf * TILE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            gmm_f16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = wn * 128 + i * 32 + frow;
                a[i]         = *(const gmm_f16x8*)(abase + rr * 128 + (((ks * 2 + fk) ^ ((rr >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rr = wt * 64 + j * 32 + frow;
                b[j]         = *(const gmm_f16x8*)(lds + rr * 128 + (((ks * 2 + fk) ^ ((rr >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        gmm_screen_epilogue(acc, g_na, g_cabs, nxv, qv, g_masks, t0 + wt * 64, r * 16 + wn * 8, lane, d.Mpad16);
    }
}

// The same screen with a wave per 32 frames x all 256 slots of a tile and ONE 32-slot block (two mixtures) in flight at a time:
// four MFMAs, then that block's reduction, so the matrix pipe works on block i + 1 while the vector pipe reduces block i, with
// 16 accumulator registers live instead of 128 -- two workgroups fit a CU (the frame tile's fragments move to registers and
// its LDS region becomes the second slot-tile buffer: 64 KB).  The slot table of this kernel (d_scr_A2) stores slot s of
// mixture 2b + h of a tile in row b*32 + (s>>2)*8 + h*4 + (s&3): in the 32x32 accumulator layout lane half h then holds all
// 16 slots of mixture 2b + h for its frame, in slot order -- the minimum and the mask need no cross-lane traffic, and one
// exchange with the partner lane per tile assembles a frame's 16 masks (32 bytes, 16 per lane).
__global__ __launch_bounds__(512, 2) void gmm_screen_rows_kernel(const _Float16* __restrict__ g_A, const _Float16* __restrict__ g_X,
                                                                const float* __restrict__ g_p1, const float* __restrict__ g_p2,
                                                                const float* __restrict__ g_nx, const float* __restrict__ g_q,
                                                                uint16_t* __restrict__ g_masks, int n_tiles_r, int r_split, int Mpad16) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int TILE = 32 * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_t = blockIdx.x / r_split, part = blockIdx.x % r_split;
    const int per = (n_tiles_r + r_split - 1) / r_split;
    const int r_begin = part * per, r_end = min(n_tiles_r, r_begin + per);
    if (r_begin >= r_end)
        return;
    const int t0 = tile_t * 256;
    const int xr = ((wave & 1) * 4 + (lane >> 4)) & 7;
    const size_t off = (size_t)(wave * 8 + (lane >> 3)) * 64 + (((lane & 7) ^ xr) << 3);
    auto load_tile = [&](const _Float16* base, char* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(base + off + (size_t)i * 64 * 64),
                                             (__attribute__((address_space(3))) void*)(dst + wave * 1024 + i * 8 * 1024), 16, 0, 0);
    };
    load_tile(g_X + (size_t)t0 * 64, lds + TILE);
    load_tile(g_A + (size_t)r_begin * 256 * 64, lds);
    const int   frow = lane & 31, fk = lane >> 5;
    const int   t    = t0 + wave * 32 + frow;
    const float nx = g_nx[t], q = g_q[t];
    const bool  all = !(nx < __builtin_inff());  // operand row did not fit f16: keep every slot
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    gmm_f16x8 bx[4];
    {
        const int rr = wave * 32 + frow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            bx[ks] = *(const gmm_f16x8*)(lds + TILE + rr * 128 + (((ks * 2 + fk) ^ ((rr >> 1) & 7)) << 4));
    }
    float p1n[8], p2n[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        p1n[i] = g_p1[r_begin * 16 + i * 2 + fk];
        p2n[i] = g_p2[r_begin * 16 + i * 2 + fk];
    }
    for (int r = r_begin; r < r_end; ++r) {
        const int buf = (r - r_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // my share of tile r is in LDS (and my frame fragments are read)
        __builtin_amdgcn_s_barrier();                                // ... everybody's, and the other buffer is free
        if (r + 1 < r_end)
            load_tile(g_A + (size_t)(r + 1) * 256 * 64, lds + (buf ^ 1) * TILE);
        float p1v[8], p2v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            p1v[i] = p1n[i];
            p2v[i] = p2n[i];
        }
        if (r + 1 < r_end) {  // the next tile's per-mixture threshold terms: a round trip to L2 that must not sit in front of
                              // the first block's reduction
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                p1n[i] = g_p1[(r + 1) * 16 + i * 2 + fk];
                p2n[i] = g_p2[(r + 1) * 16 + i * 2 + fk];
            }
        }
        const char* abase = lds + buf * TILE;
        unsigned    P[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int  rr = i * 32 + frow;
            gmm_f32x16 c;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                c[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const gmm_f16x8 a = *(const gmm_f16x8*)(abase + rr * 128 + (((ks * 2 + fk) ^ ((rr >> 1) & 7)) << 4));
                c                 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bx[ks], c, 0, 0, 0);
            }
            float mn = min3_first(c[0], c[1], c[2]);
            mn       = min3_raw(mn, c[3], c[4]);
            mn       = min3_raw(mn, c[5], c[6]);
            mn       = min3_raw(mn, c[7], c[8]);
            mn       = min3_raw(mn, c[9], c[10]);
            mn       = min3_raw(mn, c[11], c[12]);
            mn       = min3_raw(mn, c[13], c[14]);
            mn       = min3_raw(mn, c[15], c[15]);
            // tau as in gmm_screen_epilogue: p1 = 2.05 ra + 1.3e-4 sqrtK, p2 = 1.3e-4 sqrtK na + 1.6e-5 cabs, q per frame
            const float thr = mn + fmaf(nx, p1v[i], fmaf(fabsf(mn), 1.6e-5f, p2v[i] + q)) + 1e-30f;
            unsigned    bits = 0;  // bit k = "slot k is above the threshold", filled from the top down
#pragma unroll
            for (int e = 14; e >= 0; e -= 2) {
                const gmm_pk2 dd = gmm_pk2{thr, thr} - gmm_pk2{c[e], c[e + 1]};
                bits             = __builtin_amdgcn_alignbit(bits, __float_as_uint(dd.y), 31);  // (bits << 1) | sign(thr - g)
                bits             = __builtin_amdgcn_alignbit(bits, __float_as_uint(dd.x), 31);
            }
            const unsigned m16 = all ? 0xffffu : (~bits & 0xffffu);
            P[i >> 1] |= m16 << (16 * (i & 1));
        }
        // a frame's 16 masks: this lane holds mixtures i*2 + fk, the partner lane the others; lane half 0 writes masks 0..7,
        // half 1 masks 8..15, so each needs two of the partner's packed pairs
        const unsigned qa = (unsigned)__shfl_xor((int)(fk ? P[0] : P[2]), 32, 64);
        const unsigned qb = (unsigned)__shfl_xor((int)(fk ? P[1] : P[3]), 32, 64);
        const unsigned pa = fk ? P[2] : P[0], pb = fk ? P[3] : P[1];
        uint4          w;
        if (fk == 0) {  // mine are the even mixtures (low halves)
            w.x = (pa & 0xffffu) | (qa << 16);
            w.y = (pa >> 16) | (qa & 0xffff0000u);
            w.z = (pb & 0xffffu) | (qb << 16);
            w.w = (pb >> 16) | (qb & 0xffff0000u);
        }
        else {
            w.x = (qa & 0xffffu) | (pa << 16);
            w.y = (qa >> 16) | (pa & 0xffff0000u);
            w.z = (qb & 0xffffu) | (pb << 16);
            w.w = (qb >> 16) | (pb & 0xffff0000u);
        }
        *(uint4*)(g_masks + (size_t)t * Mpad16 + r * 16 + fk * 8) = w;
    }
}

// survivor rows per lane and pass of the exact stage: 28 bytes = 7 dwords, so consecutive lanes start in different banks;
// per-density covariances at DIM >= 64 leave room for one dword only (more passes for the rare frames that need them)
__host__ __device__ constexpr int gmm_list_cap(int dim, bool pooled) {
    return (pooled || dim < 64) ? 28 : 4;
}

__host__ __device__ constexpr int gmm_exact_ld(int dim, bool pooled) {
    return (pooled || dim < 64) ? 4 * (((dim + 3) / 4) | 1) : ((dim + 3) & ~3);
}

// exact evaluation of the surviving slots: thread = frame (features in registers), workgroup = 16 mixtures x 256 frames,
// the 256 slot means (and 1/sigma rows when not pooled) of the tile in LDS, rows padded to DIM + 1 floats
template<int DIM, bool POOLED, bool FMA = false>
__global__ __launch_bounds__(256) void gmm_screen_exact_kernel(const float* __restrict__ g_feats, const uint16_t* __restrict__ g_masks,
                                                              const uint32_t* __restrict__ g_mix_off, const uint32_t* __restrict__ g_k_mean,
                                                              const uint32_t* __restrict__ g_k_cov, const double* __restrict__ g_k_c64,
                                                              const float* __restrict__ g_means, const float* __restrict__ g_isr,
                                                              float* __restrict__ g_scores, uint32_t* __restrict__ g_best, int T, int n_mix,
                                                              int Mpad16, float* __restrict__ g_part_min, unsigned* __restrict__ g_part_idx,
                                                              int part_ld, int FG) {
    // rows are 16-byte aligned for ds_read_b128 (the survivor walk was co-limited by LDS issue with 8-byte reads) and an ODD
    // number of 16-byte slots long, so that consecutive rows start in different slots: 44 floats for DIM = 40
    // (per-density covariances at DIM = 64 only fit unpadded)
    constexpr int LD = gmm_exact_ld(DIM, POOLED);
    constexpr int kListCap = gmm_list_cap(DIM, POOLED);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* s_mu = (float*)lds;                    // [256][LD]
    float* s_is = s_mu + 256 * LD;                // [256][LD] (per-density covariance only)
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * 16;
    // slot -> mean / covariance row (one slot per thread), then a cooperative copy: 256 rows x DIM floats, consecutive
    // threads on consecutive floats of a row
    // plus the slot constants and the mixture sizes, so that the survivor loop touches LDS only
    double* s_c64 = (double*)(s_mu + (POOLED ? 1 : 2) * 256 * LD);  // [256]; LD is even, so this is 8-byte aligned
    int*    s_row = (int*)(s_c64 + 256);  // [256] mean row, [256] covariance row (-1 = empty slot), [16] densities per mixture
    float*  s_sc  = (float*)(s_row + 528);                 // [256][17] scores of the current frame group
    unsigned char* s_bd = (unsigned char*)(s_sc + 256 * 17);  // [256][20] best slot
    unsigned char* s_list = s_bd + 256 * 20;                   // [256][kListCap] survivor rows of the current pass
    {
        const int      m  = m0 + (tid >> 4), jj = tid & 15;
        const uint32_t k0 = m < n_mix ? g_mix_off[m] : 0u, k1 = m < n_mix ? g_mix_off[m + 1] : 0u;
        const bool     ok = k0 + jj < k1;
        s_row[tid]       = ok ? (int)g_k_mean[k0 + jj] : -1;
        s_row[256 + tid] = ok ? (int)g_k_cov[k0 + jj] : -1;
        s_c64[tid]       = ok ? g_k_c64[k0 + jj] : 0.0;
        if (jj == 0)
            s_row[512 + (tid >> 4)] = (int)(k1 - k0);
    }
    __syncthreads();
    for (int e = tid; e < 256 * DIM; e += 256) {
        const int r = e / DIM, i = e - r * DIM;
        const int mr = s_row[r];
        if (mr >= 0) {
            s_mu[r * LD + i] = g_means[(size_t)mr * DIM + i];
            if (!POOLED)
                s_is[r * LD + i] = g_isr[(size_t)s_row[256 + r] * DIM + i];
        }
    }
    __syncthreads();
    unsigned vm[8];  // validity of the 16 x 16 slots, in the layout of the mask words (mixture 2w low half, 2w + 1 high half)
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const unsigned lo = (1u << (unsigned)s_row[512 + 2 * w]) - 1u, hi = (1u << (unsigned)s_row[512 + 2 * w + 1]) - 1u;
        vm[w]             = (lo & 0xffffu) | (hi << 16);
    }
    // the tile serves FG groups of 256 frames
  // features and masks of the NEXT frame group are requested before this group's walk: they come from L2 / MALL (the
  // feature rows are re-read per mixture tile) and a round trip is a fifth of a walk
  float xn[DIM];
  uint4 man, mbn;
  auto  prefetch = [&](int fgn) {
      const int tn  = (blockIdx.y * FG + fgn) * 256 + tid;
      const int ttn = tn < T ? tn : T - 1;
#pragma unroll
      for (int i = 0; i < DIM; ++i)
          xn[i] = g_feats[(size_t)ttn * DIM + i];
      const uint4* mrow = (const uint4*)(g_masks + (size_t)ttn * Mpad16 + m0);
      man = mrow[0];
      mbn = mrow[1];
  };
  prefetch(0);
  for (int fg = 0; fg < FG; ++fg) {
    const int t = (blockIdx.y * FG + fg) * 256 + tid;
    if (t - tid >= T)
        break;
    float x[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i)
        x[i] = xn[i];
    const unsigned mw[8] = {man.x, man.y, man.z, man.w, mbn.x, mbn.y, mbn.z, mbn.w};
    if (fg + 1 < FG && (blockIdx.y * FG + fg + 1) * 256 < T)
        prefetch(fg + 1);
    // Every lane walks ITS OWN list of (mixture, slot) survivors: one distance per loop trip for every lane that still has
    // work, instead of a trip count of max-over-lanes per mixture (16 x ~2.2 trips become ~1.1 x 16 + spread).
    // The list is made first: the set bits of the eight mask words (static register indices, empty slots masked off with
    // the tile's validity words) become byte-sized row numbers w*32 + bit = mixture*16 + slot in LDS, kListCap per pass.
    // The walk then has no mask arithmetic and no data-dependent inner loops: four row numbers per 32-bit LDS read, the
    // mean row of the next survivor fetched while the current one is evaluated.  Frames whose operand did not fit f16 keep
    // all slots (up to 256 survivors) and simply take several passes.
    const int nm = min(16, n_mix - m0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {  // mixtures without densities keep the empty result
        s_sc[tid * 17 + q] = MaxState().result();
        s_bd[tid * 20 + q] = 0xff;
    }
    unsigned char* my_list = s_list + tid * kListCap;
    float          mua[DIM], mub[DIM];
    auto           fetch = [&](float (&dst)[DIM], int row) {
        const float* src = s_mu + (row < 0 ? 0 : row) * LD;
#pragma unroll
        for (int i = 0; i + 3 < DIM; i += 4) {
            const float4 v = *(const float4*)(src + i);
            dst[i]         = v.x;
            dst[i + 1]     = v.y;
            dst[i + 2]     = v.z;
            dst[i + 3]     = v.w;
        }
#pragma unroll
        for (int i = DIM & ~3; i < DIM; ++i)
            dst[i] = src[i];
    };
    MaxState st;
    auto     eval = [&](const float (&mu)[DIM], int row, int next_row) {
        const float* is   = POOLED ? g_isr : s_is + row * LD;
        const float  dist = gmm_distance_pk_reg<DIM, FMA>(x, mu, is);
        st.add(s_c64[row], 0.f, dist, (uint32_t)(row & 15));
        if ((next_row >> 4) != (row >> 4)) {  // last survivor of this mixture (next_row = -1 gives mixture -1)
            s_sc[tid * 17 + (row >> 4)] = st.result();
            s_bd[tid * 20 + (row >> 4)] = (unsigned char)st.idx;  // 0..15, or 0xff for "no density" (idx = 0xffffffff)
            st                          = MaxState();
        }
    };
    for (int base = 0;; base += kListCap) {
        int idx = 0, peek = -1;  // survivors seen so far; the first row behind this pass (decides the last flush of the pass)
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            unsigned bits = mw[w] & vm[w];
            while (bits) {
                const int bpos = __ffs((int)bits) - 1;
                bits &= bits - 1;
                const int rel = idx - base;
                if (rel >= 0 && rel < kListCap)
                    my_list[rel] = (unsigned char)(w * 32 + bpos);
                else if (rel == kListCap)
                    peek = w * 32 + bpos;
                ++idx;
            }
        }
        const int n_here = min(max(idx - base, 0), kListCap);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned q4 = *(const unsigned*)my_list;
        fetch(mua, n_here > 0 ? (int)(q4 & 0xffu) : -1);
        for (int kb = 0; kb < n_here; kb += 4) {
            const unsigned qn = *(const unsigned*)(my_list + ((kb + 4 < kListCap) ? kb + 4 : 0));  // the next four row numbers
            const int      r0 = (int)(q4 & 0xffu), r1 = (int)((q4 >> 8) & 0xffu), r2 = (int)((q4 >> 16) & 0xffu), r3 = (int)(q4 >> 24);
            const int      rn = (int)(qn & 0xffu);
            const bool     v1 = kb + 1 < n_here, v2 = kb + 2 < n_here, v3 = kb + 3 < n_here, v4 = kb + 4 < n_here;
            fetch(mub, v1 ? r1 : -1);
            eval(mua, r0, v1 ? r1 : peek);
            if (v1) {
                fetch(mua, v2 ? r2 : -1);
                eval(mub, r1, v2 ? r2 : peek);
            }
            if (v2) {
                fetch(mub, v3 ? r3 : -1);
                eval(mua, r2, v3 ? r3 : peek);
            }
            if (v3) {
                fetch(mua, v4 ? rn : -1);
                eval(mub, r3, v4 ? rn : peek);
            }
            q4 = qn;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!__any(idx > base + kListCap))  // wave-uniform: another pass only for the frames that keep every slot
            break;
    }
    // The result tile leaves through LDS: four adjacent lanes write the 64 contiguous bytes of one frame in ONE instruction.
    // (Per-lane 4-byte stores at a 40 KB stride cost more than the whole evaluation.)  Every wave transposes and stores ITS OWN
    // 64 frames, so the frame groups need no workgroup barrier: LDS operations of a wave execute in order.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (g_part_min) {  // fused statistics: best state of this tile per frame (ascending state, strict '<': first minimum)
        const int tg = (blockIdx.y * FG + fg) * 256 + tid;
        float     bm = 3.402823466e+38f;
        unsigned  bi = 0xffffffffu;
        for (int q = 0; q < nm; ++q) {
            const float v = s_sc[tid * 17 + q];
            if (v < bm) {
                bm = v;
                bi = (unsigned)(m0 + q);
            }
        }
        if (tg < T) {
            g_part_min[(size_t)blockIdx.x * part_ld + tg] = bm;
            g_part_idx[(size_t)blockIdx.x * part_ld + tg] = bi;
        }
    }
    const int  tb   = (blockIdx.y * FG + fg) * 256;
    const bool wide = nm == 16 && (n_mix & 3) == 0 && ((uintptr_t)g_scores & 15) == 0 && (!g_best || ((uintptr_t)g_best & 15) == 0);
    const int  w0   = (tid >> 6) * 64;  // first frame of this wave inside the group
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int e  = q4 * 64 + (tid & 63);           // 256 (frame, 16-byte chunk) pairs per wave
        const int fr = w0 + (e >> 2), c4 = (e & 3) * 4, tg = tb + fr;
        if (tg >= T)
            continue;
        const float*         ps = s_sc + fr * 17 + c4;
        const unsigned char* pb = s_bd + fr * 20 + c4;
        float*               gs = g_scores + (size_t)tg * n_mix + m0 + c4;
        uint32_t*            gb = g_best ? g_best + (size_t)tg * n_mix + m0 + c4 : nullptr;
        if (wide) {
            *(float4*)gs = make_float4(ps[0], ps[1], ps[2], ps[3]);
            if (gb)
                *(uint4*)gb = make_uint4(pb[0] == 0xff ? 0xffffffffu : pb[0], pb[1] == 0xff ? 0xffffffffu : pb[1],
                                         pb[2] == 0xff ? 0xffffffffu : pb[2], pb[3] == 0xff ? 0xffffffffu : pb[3]);
        }
        else {
            for (int q = 0; q < 4 && c4 + q < nm; ++q) {
                gs[q] = ps[q];
                if (gb)
                    gb[q] = pb[q] == 0xff ? 0xffffffffu : pb[q];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

struct GmmCombineDims {
    int T, Tpad, n_mix, mix_tile;
};

template<class State>
__global__ __launch_bounds__(256) void gmm_combine_kernel(const float* __restrict__ g_dist, float* __restrict__ g_scores,
                                                         uint32_t* __restrict__ g_best, const uint32_t* __restrict__ g_mix_off,
                                                         const uint32_t* __restrict__ g_k_dens, const double* __restrict__ g_k_c64,
                                                         const float* __restrict__ g_k_c32, GmmCombineDims dims) {
    struct {
        const float* __restrict__ dist; float* __restrict__ scores; uint32_t* __restrict__ best;
        const uint32_t* __restrict__ mix_off; const uint32_t* __restrict__ k_dens; const double* __restrict__ k_c64;
        const float* __restrict__ k_c32; int T, Tpad, n_mix, mix_tile;
    } p = {g_dist, g_scores, g_best, g_mix_off, g_k_dens, g_k_c64, g_k_c32, dims.T, dims.Tpad, dims.n_mix, dims.mix_tile};
    const int  lane = threadIdx.x & 63;
    const int  wave = threadIdx.x >> 6;
    const int  t    = (blockIdx.y * 4 + wave) * 64 + lane;
    const bool live = t < p.T;
    const int  tt   = live ? t : (p.T - 1);
    const int  m0   = blockIdx.x * p.mix_tile;
    const int  m1   = min(m0 + p.mix_tile, p.n_mix);
    for (int m = m0; m < m1; ++m) {
        const uint32_t k0 = p.mix_off[m], k1 = p.mix_off[m + 1];
        State          st;
        for (uint32_t k = k0; k < k1; ++k) {
            float dist = p.dist[(size_t)p.k_dens[k] * p.Tpad + tt];
            st.add(p.k_c64[k], p.k_c32[k], dist, k - k0);
        }
        if (live) {
            p.scores[(size_t)t * p.n_mix + m] = st.result();
            if (p.best)
                p.best[(size_t)t * p.n_mix + m] = st.idx;
        }
    }
}

}  // namespace amx

// ------------------------------------------------------------------------------------ ABI

struct amx_gmm {
    amx_ctx* ctx = nullptr;
    int      dim = 0, n_mix = 0, n_dens = 0, n_mean = 0, n_cov = 0;
    size_t   nk = 0;
    // host copies of the prepared tables (amx_gmm_tables)
    std::vector<float>    m2lw, isr, lognorm;
    std::vector<uint32_t> mix_off, h_k_dens, h_d_mean, h_d_cov;  // topology (accumulator files)
    // device
    uint32_t *d_mix_off = nullptr, *d_k_mean = nullptr, *d_k_cov = nullptr, *d_k_dens = nullptr;
    float *   d_means_t = nullptr, *d_isr_t = nullptr;  // same models: means / inverse deviations in list order, [dim][Kpad] (gmm_dist_list_kernel)
    uint32_t* d_dens_pos = nullptr;  // shared-list models whose list names every density at most once: density -> list position (~0: not listed)
    uint32_t *d_d_mean = nullptr, *d_d_cov = nullptr;
    double*   d_k_c64 = nullptr;
    float *   d_k_c32 = nullptr, *d_means = nullptr, *d_isr = nullptr;
    // batch-float mode tables (pooled covariance only)
    float *   d_smeans = nullptr, *d_k_const = nullptr, *d_isr0 = nullptr;
    bool      pooled = false;
    bool      tied    = false;  // use the two-stage path
    bool      uniform = false;  // tied AND every mixture lists the same densities: lane = mixture combine
    int       K = 0, mix_pad = 0;
    float*    d_m2lw_t = nullptr;  // [K][mix_pad]
    float *   d_ahat_t = nullptr, *d_amax = nullptr;  // screen tables: fl32(m2lw + logNorm) [K][mix_pad], max_k |.| [mix_pad]
    // pruned path (gmm_tied.hip): per-tile minima of a^, per-call workspace, survivor statistics of earlier calls
    float*              d_amin       = nullptr;  // [mix_pad / 64 + 1][Kpad]
    unsigned short*     d_aup        = nullptr;  // [K][mix_pad] bf16 image of a^, rounded up (bounds only)
    void*               d_tied_ws    = nullptr;
    size_t              tied_ws_cap  = 0;
    unsigned long long* d_tied_surv  = nullptr;  // [256] survivors (density, frame, tile) of the calls so far, spread over 256 counters; [256] = triples examined
    unsigned long long* h_tied_surv  = nullptr;  // pinned host copy, refreshed asynchronously after every 8th pruned call (tied_publish)
    unsigned            tied_copy_tick = 0;      // pruned calls since the handle was made
    bool                tied_capturing = false;
    bool                tied_keys_clean = false;  // the workspace's near keys are in their empty state (gmm_dist_list_kernel's atomics start from it)  // the pruned launches are being recorded: the copy stays outside the graph
    unsigned long long  tied_seen    = 0;        // survivors / examined triples in the host copy at the previous decision
    unsigned long long  tied_triples = 0;
    int                 tied_dense_calls = 0;    // > 0: stay on gmm_tied_tile_kernel for that many calls, then probe again
    unsigned long long  tied_rep_seen = 0, tied_rep_triples = 0;  // amx_gmm_screen_counts: counter value / triples at the last report
    int                 tied_forced = -1;        // set around a nested call: 1 = pruned path, 0 = dense kernel, -1 = decide
    double *  d_ln64 = nullptr, *d_dist64 = nullptr;
    float*    d_ln32 = nullptr;
    size_t    dist64_cap = 0;
    float*    d_dist  = nullptr;
    size_t    dist_floats = 0;
    // MFMA screen (private-density models, <= 16 densities per mixture; see gmm_screen_kernel)
    bool      screen = false;
    int       scr_Kp = 0, scr_Rpad = 0, scr_Mpad16 = 0;
    float     scr_rmax2 = 0.f, scr_na_all = 0.f;
    _Float16* d_scr_A = nullptr;
    _Float16* d_scr_A2 = nullptr;  // K = 64 only: the slot rows in gmm_screen_rows_kernel's order
    float *   d_scr_c = nullptr, *d_scr_na = nullptr, *d_scr_cabs = nullptr;
    // per-call workspace of the screen
    _Float16* d_scr_X = nullptr;
    float *   d_scr_nx = nullptr, *d_scr_q = nullptr;
    uint16_t* d_scr_masks = nullptr;
    int       scr_cap_T = 0;
    // staging buffers of the host-buffer entry point amx_gmm_score
    float *   d_host_f = nullptr, *d_host_s = nullptr;
    uint32_t* d_host_b = nullptr;
    size_t    host_f_cap = 0, host_s_cap = 0, host_b_cap = 0;
    void*     simd = nullptr;        // SIMD-diagonal-maximum tables and workspaces (gmm_simd.hip), built on first use
    std::vector<float>  h_means, h_vars;   // host copies of the model for that lazy build
    std::vector<double> h_logw;
    double    mws = 1.0, gsc = 1.0;
    int       simd_status = 0;       // 0 = not built yet, 1 = built, < 0 = amx_status of the failed build
    // preselection-batch-float (gmm_presel.hip): density clustering, built on first use / when its parameters change
    void*     presel = nullptr;
    int       presel_clusters = 256, presel_select = 32, presel_iterations = 5;  // Mm/DensityClustering.cc:21-35 defaults
    float     presel_backoff = 40000.f;
    std::vector<uint32_t> h_k_mean;
    std::vector<float>    h_smeans;
    // small-batch passes of the screened scorer on unchanged device buffers (the decoder's ring buffer), replayed as HIP graphs
    struct GraphKey {
        const void *feats, *scores, *best;
        hipStream_t stream;
        int         T;
        bool operator<(const GraphKey& o) const {
            return std::tie(feats, scores, best, stream, T) < std::tie(o.feats, o.scores, o.best, o.stream, o.T);
        }
    };
    std::map<GraphKey, hipGraphExec_t> graphs;
    int                                use_graphs = 1;
    // amx_gmm_model.tuning (A/B runs, tests)
    int         tune_screen = 1, tune_fused = 1, tune_screen_all = 0, tune_tied_prune = -1, tune_chunk = 65536, tune_fused_waves = 0, tune_fr = 8,
                tune_simd_mfma = 1, tune_dist_list = 1, tune_near_fused = 1, tune_fused_pack = 1;
    std::string tune_screen_kernel = "rows";
    // amx_gmm_model.tuning contract=fma: the distance's `sum += df * df` as one fused multiply-add = the reference's default build
    // (-march=native on an FMA host); off (default) = the reference built with -DMARCH=x86-64.  Not a speed switch: it selects WHICH
    // build of RASR the scores are bit-identical to.
    bool        contract_fma = false;
    void*     d_fus_rec = nullptr;   // tile records of gmm_fused_kernel (pooled covariance, dim <= 40)
    uint32_t* d_best32   = nullptr;  // u32 workspace of amx_gmm_score_stats_u8_dev on paths without a byte-writing kernel
    size_t    best32_cap = 0;
    unsigned long long* d_fus_surv = nullptr;  // [256] partial counts of the densities evaluated exactly (amx_gmm_screen_counts): one
                                               // address for every wave's atomicAdd cost a quarter of a 256-frame pass
    size_t    fus_rec_bytes = 0;
    bool      count_survivors = false;
    unsigned long long fus_pairs = 0;
    float*    d_scr_pmin = nullptr;  // fused statistics: per-tile arg-min partials
    unsigned* d_scr_pidx = nullptr;
    size_t    scr_part_cap = 0;
};

namespace {

template<class T>
int gupload(T** dst, const T* src, size_t n) {
    AMX_HIP(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
    if (n)
        AMX_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return AMX_OK;
}

bool screen_dim_supported(int d) {
    switch (d) {
        case 16: case 24: case 32: case 33: case 39: case 40: case 45: case 48: case 64: return true;
        default: return false;
    }
}

extern "C" int   amx_internal_gmm_presel_create(amx_ctx* ctx, int dim, size_t nk, const uint32_t* k_mean_host, const float* smeans_host,
                                                const float* d_smeans, const uint32_t* d_k_mean, int n_clusters, int n_select, int iterations,
                                                float backoff, int contract_fma, void** out);
extern "C" void  amx_internal_gmm_presel_destroy(void* p);
extern "C" int   amx_internal_gmm_presel_info(const void* p, int* n_clusters, uint32_t* cluster_of, float* cluster_means);
extern "C" int   amx_internal_gmm_presel_score(void* p, amx_ctx* ctx, const float* feats_dev, int T, float* scores_dev, const uint32_t* d_mix_off,
                                               const uint32_t* d_k_mean, const float* d_k_const, const float* d_smeans, const float* d_isr0,
                                               int n_mix);
extern "C" int   amx_internal_gmm_simd_create(const amx_gmm_model* m, int contract_fma, void** out, float* scaling_out);
extern "C" void  amx_internal_gmm_simd_destroy(void* p);
extern "C" float amx_internal_gmm_simd_scaling(const void* p);
extern "C" int   amx_internal_gmm_simd_score(void* p, amx_ctx* ctx, int variant, const float* feats_dev, int T, float* scores_dev, uint32_t* best_dev);
extern "C" int   amx_internal_gmm_simd_presel_build(void* p, amx_ctx* ctx, int n_clusters, int n_select, int iterations);
extern "C" int   amx_internal_gmm_simd_presel_info(const void* p, int* n_clusters, uint32_t* cluster_of, float* cluster_means);
extern "C" int   amx_internal_gmm_simd_presel_score(void* p, amx_ctx* ctx, const float* feats_dev, int T, float* scores_dev);

extern "C" int    amx_internal_gmm_tied_create(int K, int n_mix, int mix_pad, const float* ahat_t_host, float** d_amin, unsigned short** d_aup);
extern "C" size_t amx_internal_gmm_tied_workspace(int K, int T, int mix_pad);
extern "C" int    amx_internal_gmm_tied_score(amx_ctx* ctx, const float* dist_dev, const uint32_t* k_dens_dev, int K, int T, int Tpad, int n_mix,
                                              int mix_pad, const unsigned short* aup, const float* amax, const float* m2lw_t, const float* ahat_t,
                                              const double* ln64, const float* ln32, const float* amin, void* workspace, float* scores,
                                              uint32_t* best, unsigned long long* survivors_dev, int dt_written, int near_written);
extern "C" float* amx_internal_gmm_tied_dt(void* workspace, int K, int T, int have_positions);
extern "C" unsigned long long* amx_internal_gmm_tied_near(void* workspace);
extern "C" int                 amx_internal_gmm_tied_near_init(amx_ctx* ctx, void* workspace);
extern "C" int amx_internal_gmm_fused_supported(int dim, int pooled, int Kp);
extern "C" int amx_internal_gmm_fused_create(int dim, int n_mix, int n_tiles, const void* A2_host, const uint32_t* mix_off, const uint32_t* k_mean,
                                             const double* c64, const float* means, const float* p1, const float* p2, void** rec_dev,
                                             size_t* rec_bytes);
extern "C" int amx_internal_gmm_fused_split(int n_cu, int Tpad, int n_tiles, int forced_waves);
extern "C" int amx_internal_gmm_fused_score(amx_ctx* ctx, int dim, const void* rec_dev, const float* isr_dev, const float* feats, const void* X,
                                            const float* nx, const float* q, int T, int Tpad, int n_mix, int n_tiles, int split, float* scores,
                                            uint32_t* best, float* pmin, unsigned* pidx, int part_ld, unsigned long long* survivors, int forced_waves,
                                            int best_bytes, int contract_fma, float na_all);
extern "C" int amx_internal_gmm_fused_waves(int Tpad, int forced_waves);

// maximum approximation through the MFMA screen (see gmm_screen_kernel); frames in chunks that bound the mask workspace
extern "C" int amx_internal_best_state_reduce(amx_ctx*, const float*, const unsigned*, int, int, int, uint32_t*, unsigned long long*, double*);

// best_bytes = 1 (fused path only, see fused_bytes_ok): best_dev is a byte matrix [T x n_mix] (amx_gmm_score_stats_u8_dev)
int score_screened(amx_gmm* h, const float* feats_dev, int T, float* scores_dev, uint32_t* best_dev, bool stats, uint32_t* best_state_dev,
                   unsigned long long* counts_dev, double* score_sum_dev, int best_bytes = 4) {
    hipStream_t st = h->ctx->stream;
    const int   chunk = h->tune_chunk;  // frames per pass: the workspace (survivor masks, 2 B per frame and mixture slot) grows to what a call needs
    // one fused kernel (gmm_fused.hip) where its tile records exist; tuning fused=0 keeps the two-kernel path (A/B runs, tests)
    const bool fused = h->d_fus_rec && h->tune_fused && !h->tune_screen_all;
    AMX_REQUIRE(best_bytes == 4 || (best_bytes == 1 && fused), AMX_ERR_STATE, "score_screened: byte-sized best densities exist on the fused path only");
    for (int t0 = 0; t0 < T; t0 += chunk) {
        const int Tc = std::min(chunk, T - t0), Tpad = (Tc + 255) / 256 * 256;
        if (Tpad > h->scr_cap_T || (!fused && !h->d_scr_masks)) {
            for (auto& kv : h->graphs)  // captured passes hold the old workspace addresses: drop them before the buffers move
                if (kv.second)
                    hipGraphExecDestroy(kv.second);
            h->graphs.clear();
            hipFree(h->d_scr_X);
            hipFree(h->d_scr_nx);
            hipFree(h->d_scr_q);
            hipFree(h->d_scr_masks);
            h->d_scr_X = nullptr;
            h->d_scr_nx = h->d_scr_q = nullptr;
            h->d_scr_masks = nullptr;
            h->scr_cap_T = 0;
            AMX_HIP(hipMalloc((void**)&h->d_scr_X, (size_t)Tpad * h->scr_Kp * sizeof(_Float16)));
            AMX_HIP(hipMalloc((void**)&h->d_scr_nx, (size_t)Tpad * 4));
            AMX_HIP(hipMalloc((void**)&h->d_scr_q, (size_t)Tpad * 4));
            if (!fused)
                AMX_HIP(hipMalloc((void**)&h->d_scr_masks, (size_t)Tpad * h->scr_Mpad16 * 2));
            h->scr_cap_T = Tpad;
        }
        const float*       x = feats_dev + (size_t)t0 * h->dim;
        amx::GmmScreenDims d{Tc, Tpad, h->dim, h->scr_Kp, h->scr_Mpad16, h->pooled ? 1 : 0, h->scr_rmax2,
                             std::sqrt((float)(h->pooled ? h->dim : 2 * h->dim)), h->scr_na_all};
        // the fused kernel packs its own operand rows (fused_pack=1, default; the wave-specialised lab kernel reads packed rows)
        const bool own_pack = fused && h->tune_fused_pack && h->tune_fused_waves != 13;
        if (!own_pack) {
            amx::ScopedKernelTimer timer(h->ctx, "gmm_screen_pack");
            hipLaunchKernelGGL(amx::gmm_screen_pack_kernel, dim3(Tpad / 4), dim3(256), 0, st, x, h->d_isr, h->d_scr_X, h->d_scr_nx, h->d_scr_q, d);
        }
        if (fused) {
            const int n_tiles = h->scr_Rpad / 256;
            const int split   = amx_internal_gmm_fused_split(h->ctx->n_cu, Tpad, n_tiles, h->tune_fused_waves);
            float*    pmin    = nullptr;
            unsigned* pidx    = nullptr;
            if (stats) {
                const size_t need = (size_t)split * Tpad;
                if (need > h->scr_part_cap) {
                    hipFree(h->d_scr_pmin);
                    hipFree(h->d_scr_pidx);
                    h->d_scr_pmin   = nullptr;
                    h->d_scr_pidx   = nullptr;
                    h->scr_part_cap = 0;
                    AMX_HIP(hipMalloc((void**)&h->d_scr_pmin, need * 4));
                    AMX_HIP(hipMalloc((void**)&h->d_scr_pidx, need * 4));
                    h->scr_part_cap = need;
                }
                pmin = h->d_scr_pmin;
                pidx = h->d_scr_pidx;
            }
            {
                amx::ScopedKernelTimer timer(h->ctx, "gmm");
                int r = amx_internal_gmm_fused_score(h->ctx, h->dim, h->d_fus_rec, h->d_isr, x, own_pack ? nullptr : h->d_scr_X, h->d_scr_nx, h->d_scr_q, Tc, Tpad,
                                                     h->n_mix, n_tiles, split, scores_dev + (size_t)t0 * h->n_mix,
                                                     best_dev ? (uint32_t*)((char*)best_dev + (size_t)t0 * h->n_mix * best_bytes) : nullptr, pmin,
                                                     pidx, Tpad, h->count_survivors ? h->d_fus_surv : nullptr, h->tune_fused_waves, best_bytes,
                                                     h->contract_fma ? 1 : 0, h->scr_na_all);
                if (r != AMX_OK)
                    return r;
                if (h->count_survivors)
                    h->fus_pairs += (unsigned long long)Tc * (unsigned long long)h->n_mix;
            }
            if (stats) {
                amx::ScopedKernelTimer timer(h->ctx, "stats");
                int r = amx_internal_best_state_reduce(h->ctx, pmin, pidx, split, Tpad, Tc, best_state_dev ? best_state_dev + t0 : nullptr,
                                                       counts_dev, score_sum_dev);
                if (r != AMX_OK)
                    return r;
            }
            continue;
        }
        {
            amx::ScopedKernelTimer timer(h->ctx, "gmm_screen");
            const int ntr = h->scr_Rpad / 256, ntt = Tpad / 256;
            const char* variant = h->tune_screen_kernel.c_str();  // rows (default) | persist | simple
            if (h->scr_Kp == 64 && !strcmp(variant, "rows")) {
                auto      k   = amx::gmm_screen_rows_kernel;
                const int lds = 2 * 32 * 1024;
                hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                // two workgroups per CU, each a frame tile x a contiguous range of slot tiles
                const int split = std::max(1, std::min(ntr, (2 * std::max(h->ctx->n_cu, 8) + ntt - 1) / ntt));
                hipLaunchKernelGGL(k, dim3(ntt * split), dim3(512), lds, st, h->d_scr_A2, h->d_scr_X, h->d_scr_na, h->d_scr_cabs, h->d_scr_nx,
                                   h->d_scr_q, h->d_scr_masks, ntr, split, h->scr_Mpad16);
            }
            else if (h->scr_Kp == 64 && strcmp(variant, "simple")) {
                auto      k   = amx::gmm_screen_persist_kernel;
                const int lds = 3 * 32 * 1024 + 2048;
                hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                // one workgroup per CU, each a frame tile x a contiguous range of slot tiles
                const int split = std::max(1, std::min(ntr, (std::max(h->ctx->n_cu, 8) + ntt - 1) / ntt));
                hipLaunchKernelGGL(k, dim3(ntt * split), dim3(512), lds, st, h->d_scr_A, h->d_scr_X, h->d_scr_c, h->d_scr_na, h->d_scr_cabs,
                                   h->d_scr_nx, h->d_scr_q, h->d_scr_masks, ntr, split, d);
            }
            else if (h->scr_Kp == 64) {
                auto k = amx::gmm_screen_kernel<1>;
                hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 + 1024);
                hipLaunchKernelGGL(k, dim3(ntr * ntt), dim3(512), 64 * 1024 + 1024, st, h->d_scr_A, h->d_scr_X, h->d_scr_c, h->d_scr_na,
                                   h->d_scr_cabs, h->d_scr_nx, h->d_scr_q, h->d_scr_masks, ntr, d);
            }
            else {
                auto k = amx::gmm_screen_kernel<2>;
                hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024 + 1024);
                hipLaunchKernelGGL(k, dim3(ntr * ntt), dim3(512), 128 * 1024 + 1024, st, h->d_scr_A, h->d_scr_X, h->d_scr_c, h->d_scr_na,
                                   h->d_scr_cabs, h->d_scr_nx, h->d_scr_q, h->d_scr_masks, ntr, d);
            }
        }
        if (h->tune_screen_all)  // debugging aid: every slot survives (the exact stage then evaluates all densities)
            hipMemsetAsync(h->d_scr_masks, 0xff, (size_t)Tpad * h->scr_Mpad16 * 2, st);
        float*    pmin = nullptr;
        unsigned* pidx = nullptr;
        if (stats) {
            const size_t need = (size_t)(h->scr_Mpad16 / 16) * Tpad;
            if (need > h->scr_part_cap) {
                hipFree(h->d_scr_pmin);
                hipFree(h->d_scr_pidx);
                h->d_scr_pmin = nullptr;
                h->d_scr_pidx = nullptr;
                h->scr_part_cap = 0;
                AMX_HIP(hipMalloc((void**)&h->d_scr_pmin, need * 4));
                AMX_HIP(hipMalloc((void**)&h->d_scr_pidx, need * 4));
                h->scr_part_cap = need;
            }
            pmin = h->d_scr_pmin;
            pidx = h->d_scr_pidx;
        }
        {
            amx::ScopedKernelTimer timer(h->ctx, "gmm");
            // FG frame groups of 256 share one slot tile in LDS; enough of them that the tile load is amortised, few enough that
            // ~2500 workgroups remain (measured: 8 at 8192 frames, 32 at 64 k frames)
            const int FG = std::max(1, std::min(32, (int)((long)(Tpad / 256) * (h->scr_Mpad16 / 16) / 2500)));
            dim3      grid(h->scr_Mpad16 / 16, (Tpad / 256 + FG - 1) / FG);
            float*    sc = scores_dev + (size_t)t0 * h->n_mix;
            uint32_t* bd = best_dev ? best_dev + (size_t)t0 * h->n_mix : nullptr;
#define AMX_EXACT(D)                                                                                                                \
    case D: {                                                                                                                       \
        const size_t lds = (size_t)256 * amx::gmm_exact_ld(D, h->pooled) * 4 * (h->pooled ? 1 : 2) + 2048 + 2112 + 256 * 17 * 4 + 256 * 20 + 256 * amx::gmm_list_cap(D, h->pooled); \
        if (h->pooled) {                                                                                                            \
            auto k = h->contract_fma ? amx::gmm_screen_exact_kernel<D, true, true> : amx::gmm_screen_exact_kernel<D, true, false>;  \
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
            hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, h->d_scr_masks, h->d_mix_off, h->d_k_mean, h->d_k_cov, h->d_k_c64,   \
                               h->d_means, h->d_isr, sc, bd, Tc, h->n_mix, h->scr_Mpad16, pmin, pidx, Tpad, FG);                                          \
        }                                                                                                                           \
        else {                                                                                                                      \
            auto k = h->contract_fma ? amx::gmm_screen_exact_kernel<D, false, true> : amx::gmm_screen_exact_kernel<D, false, false>; \
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
            hipLaunchKernelGGL(k, grid, dim3(256), lds, st, x, h->d_scr_masks, h->d_mix_off, h->d_k_mean, h->d_k_cov, h->d_k_c64,   \
                               h->d_means, h->d_isr, sc, bd, Tc, h->n_mix, h->scr_Mpad16, pmin, pidx, Tpad, FG);                                          \
        }                                                                                                                           \
    } break;
            switch (h->dim) {
                AMX_EXACT(16)
                AMX_EXACT(24)
                AMX_EXACT(32)
                AMX_EXACT(33)
                AMX_EXACT(39)
                AMX_EXACT(40)
                AMX_EXACT(45)
                AMX_EXACT(48)
                AMX_EXACT(64)
                default: break;
            }
#undef AMX_EXACT
        }
        AMX_HIP(hipGetLastError());
        if (stats) {
            amx::ScopedKernelTimer timer(h->ctx, "stats");
            int r = amx_internal_best_state_reduce(h->ctx, pmin, pidx, h->scr_Mpad16 / 16, Tpad, Tc, best_state_dev ? best_state_dev + t0 : nullptr,
                                                   counts_dev, score_sum_dev);
            if (r != AMX_OK)
                return r;
        }
    }
    return AMX_OK;
}

template<class State>
int launch_direct(amx_gmm* h, const amx::GmmParams& p, dim3 grid) {
    hipStream_t  st  = h->ctx->stream;
    size_t       lds = 0;
    amx::GmmDims dims{p.T, p.dim, p.n_mix, p.mix_tile};
    switch (h->dim) {
#define AMX_GMM_CASE(D)                                                                                  \
    case D: {                                                                                            \
        auto k = h->contract_fma ? amx::gmm_direct_kernel<D, State, true> : amx::gmm_direct_kernel<D, State, false>; \
        hipLaunchKernelGGL(k, grid, dim3(256), 0, st, p.feats, p.scores, p.best,                          \
                               p.mix_off, p.k_mean, p.k_cov, p.k_c64, p.k_c32, p.means, p.isr, dims);       \
    } break;
        AMX_GMM_CASE(16)
        AMX_GMM_CASE(24)
        AMX_GMM_CASE(32)
        AMX_GMM_CASE(33)
        AMX_GMM_CASE(39)
        AMX_GMM_CASE(40)
        AMX_GMM_CASE(45)
        AMX_GMM_CASE(48)
        AMX_GMM_CASE(64)
#undef AMX_GMM_CASE
        default:
            lds = (size_t)4 * 64 * h->dim * sizeof(float);
            hipLaunchKernelGGL((h->contract_fma ? amx::gmm_direct_kernel<0, State, true> : amx::gmm_direct_kernel<0, State, false>), grid, dim3(256),
                               lds, st, p.feats, p.scores, p.best, p.mix_off, p.k_mean, p.k_cov, p.k_c64, p.k_c32, p.means, p.isr, dims);
    }
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int launch_dist(amx_gmm* h, const amx::GmmDistParams& p, dim3 grid, double* dist64, bool stage, float* dt = nullptr, int dt_ld = 0) {
    hipStream_t      st  = h->ctx->stream;
    size_t           lds = 0;
    amx::GmmDistDims dims{p.T, p.Tpad, p.dim, p.n_dens, p.dens_tile};
    switch (h->dim) {
#define AMX_GMM_CASE(D)                                                                                                                   \
    case D:                                                                                                                               \
        if (stage)                                                                                                                        \
            hipLaunchKernelGGL((h->contract_fma ? amx::gmm_dist_kernel<D, true, true> : amx::gmm_dist_kernel<D, true, false>), grid, dim3(256), \
                               (size_t)256 * (D + 1) * sizeof(float), st, p.feats, p.dist,                                               \
                               dist64, p.d_mean, p.d_cov, p.means, p.isr, dims, dt, (const uint32_t*)h->d_dens_pos, dt_ld);               \
        else                                                                                                                              \
            hipLaunchKernelGGL((h->contract_fma ? amx::gmm_dist_kernel<D, false, true> : amx::gmm_dist_kernel<D, false, false>), grid,    \
                               dim3(256), 0, st, p.feats, p.dist, dist64, p.d_mean, p.d_cov, p.means,                                    \
                               p.isr, dims, dt, (const uint32_t*)h->d_dens_pos, dt_ld);                                                   \
        break;
        AMX_GMM_CASE(16)
        AMX_GMM_CASE(24)
        AMX_GMM_CASE(32)
        AMX_GMM_CASE(33)
        AMX_GMM_CASE(39)
        AMX_GMM_CASE(40)
        AMX_GMM_CASE(45)
        AMX_GMM_CASE(48)
        AMX_GMM_CASE(64)
#undef AMX_GMM_CASE
        default:
            lds = (size_t)4 * 64 * h->dim * sizeof(float);
            hipLaunchKernelGGL((h->contract_fma ? amx::gmm_dist_kernel<0, false, true> : amx::gmm_dist_kernel<0, false, false>), grid, dim3(256), lds, st, p.feats, p.dist, dist64, p.d_mean, p.d_cov, p.means,
                               p.isr, dims, dt, (const uint32_t*)h->d_dens_pos, dt_ld);
    }
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

// the pruned path's frame-major distances straight from the transposed model tables; AMX_ERR_STATE: no instance for this dimension
int launch_dist_list(amx_gmm* h, const float* feats, int T, float* dt, unsigned long long* near) {
    const int Kpad = (h->K + 63) & ~63;
    // frames per wave: the 2 x dim table registers are loaded once per wave, so as many frames as still leave every SIMD two waves
    int frames = 16;
    while (frames > 2 && (long)amx::ceil_div(Kpad, 256) * 4 * amx::ceil_div(T, frames) < 8L * std::max(h->ctx->n_cu, 1))
        frames /= 2;
    if (h->tune_dist_list > 1)
        frames = h->tune_dist_list;
    const dim3 grid(amx::ceil_div(Kpad, 256), amx::ceil_div(T, frames));
    switch (h->dim) {
#define AMX_GMM_CASE(D)                                                                                                                         \
    case D:                                                                                                                                     \
        if (near)                                                                                                                               \
            hipLaunchKernelGGL((h->contract_fma ? amx::gmm_dist_list_kernel<D, true, true> : amx::gmm_dist_list_kernel<D, false, true>), grid,  \
                               dim3(256), 0, h->ctx->stream, feats, h->d_means_t, h->d_isr_t, h->K, Kpad, T, frames, dt, near);                 \
        else                                                                                                                                    \
            hipLaunchKernelGGL((h->contract_fma ? amx::gmm_dist_list_kernel<D, true, false> : amx::gmm_dist_list_kernel<D, false, false>), grid,\
                               dim3(256), 0, h->ctx->stream, feats, h->d_means_t, h->d_isr_t, h->K, Kpad, T, frames, dt, near);                 \
        break;
        AMX_GMM_CASE(16)
        AMX_GMM_CASE(24)
        AMX_GMM_CASE(32)
        AMX_GMM_CASE(33)
        AMX_GMM_CASE(39)
        AMX_GMM_CASE(40)
        AMX_GMM_CASE(45)
        AMX_GMM_CASE(48)
#undef AMX_GMM_CASE
        default:
            return AMX_ERR_STATE;
    }
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

}  // namespace

extern "C" {

int amx_gmm_create(amx_ctx* ctx, const amx_gmm_model* m, amx_gmm** out) {
    // ctx == NULL creates a host-only handle (prepared tables only; scoring returns AMX_ERR_STATE)
    AMX_REQUIRE(m && out, AMX_ERR_INVALID, "amx_gmm_create: NULL argument");
    *out = nullptr;
    AMX_REQUIRE(m->dim > 0 && m->n_mix > 0 && m->n_dens > 0 && m->n_mean > 0 && m->n_cov > 0, AMX_ERR_INVALID,
                "amx_gmm_create: empty mixture set");
    AMX_REQUIRE(m->dim <= 1024, AMX_ERR_UNSUPPORTED, "amx_gmm_create: feature dimension %d > 1024", m->dim);
    AMX_REQUIRE(m->mix_offsets && m->dens_index && m->log_weight && m->dens_mean && m->dens_cov && m->means && m->variances,
                AMX_ERR_INVALID, "amx_gmm_create: NULL table");
    const size_t nk = m->mix_offsets[m->n_mix];
    for (int i = 0; i < m->n_mix; ++i)
        AMX_REQUIRE(m->mix_offsets[i] <= m->mix_offsets[i + 1], AMX_ERR_INVALID, "amx_gmm_create: mix_offsets not monotone");
    for (size_t k = 0; k < nk; ++k)
        AMX_REQUIRE(m->dens_index[k] < (uint32_t)m->n_dens, AMX_ERR_INVALID, "amx_gmm_create: density index out of range");
    for (int d = 0; d < m->n_dens; ++d)
        AMX_REQUIRE(m->dens_mean[d] < (uint32_t)m->n_mean && m->dens_cov[d] < (uint32_t)m->n_cov, AMX_ERR_INVALID,
                    "amx_gmm_create: mean/covariance index out of range");
    // CovarianceFeatureScorerElement::checkDiagonal: require(all variances > 0)
    for (size_t i = 0; i < (size_t)m->n_cov * m->dim; ++i)
        AMX_REQUIRE(m->variances[i] > 0, AMX_ERR_INVALID, "amx_gmm_create: non-positive variance");

    amx::Tuning tune;
    if (!tune.parse(m->tuning, amx::gmm_tuning_keys, "amx_gmm_create"))
        return AMX_ERR_INVALID;
    // values are checked like keys: a typo must not silently select the default kernel (or the other arithmetic)
    int         t_screen, t_fused, t_screen_all, t_tied_prune, t_chunk, t_fused_waves, t_fr, t_simd_mfma, t_graph, t_dist_list, t_near_fused, t_fused_pack;
    std::string t_screen_kernel, t_contract;
    static const char* const screen_kernels[] = {"rows", "persist", "simple", nullptr};
    static const char* const contracts[]      = {"off", "fma", nullptr};
    const char*              who              = "amx_gmm_create";
    if (!tune.get_int("screen", 1, 0, 1, &t_screen, who) || !tune.get_int("fused", 1, 0, 1, &t_fused, who) ||
        !tune.get_int("screen_all", 0, 0, 1, &t_screen_all, who) || !tune.get_int("tied_prune", -1, -1, 1, &t_tied_prune, who) ||
        !tune.get_int("chunk", 65536, 256, 1 << 24, &t_chunk, who) || !tune.get_int("fused_waves", 0, 0, 16, &t_fused_waves, who) ||
        !tune.get_int("fr", 8, 2, 16, &t_fr, who) || !tune.get_int("simd_mfma", 1, 0, 1, &t_simd_mfma, who) ||
        !tune.get_int("graph", 0, 0, 1, &t_graph, who) || !tune.get_int("dist_list", 1, 0, 64, &t_dist_list, who) ||
        !tune.get_int("near_fused", 1, 0, 1, &t_near_fused, who) || !tune.get_int("fused_pack", 1, 0, 1, &t_fused_pack, who) || !tune.get_word("screen_kernel", "rows", screen_kernels, &t_screen_kernel, who) ||
        !tune.get_word("contract", ctx && ctx->contract == AMX_CONTRACT_FMA ? "fma" : "off", contracts, &t_contract, who))   // no key: the context's arithmetic (amx_set_contract)
        return AMX_ERR_INVALID;
    AMX_REQUIRE(t_fused_waves == 0 || t_fused_waves == 8 || t_fused_waves == 12 || t_fused_waves == 13 || t_fused_waves == 16, AMX_ERR_INVALID,
                "amx_gmm_create: tuning fused_waves=%d: expected 8 | 12 | 13 | 16", t_fused_waves);
    AMX_REQUIRE(t_fr == 2 || t_fr == 4 || t_fr == 8 || t_fr == 16, AMX_ERR_INVALID, "amx_gmm_create: tuning fr=%d: expected 2 | 4 | 8 | 16", t_fr);
    // contract=fma exists for the specialised-wave experiment's kernel neither (fused_waves=13): a measured-slower lab form
    AMX_REQUIRE(!(t_contract == "fma" && t_fused_waves == 13), AMX_ERR_UNSUPPORTED,
                "amx_gmm_create: tuning contract=fma has no fused_waves=13 kernel (the specialised-wave form exists for contract=off only)");
    amx_gmm* h = new amx_gmm;
    h->tune_screen        = t_screen;
    h->tune_fused         = t_fused;
    h->tune_screen_all    = t_screen_all;
    h->tune_tied_prune    = t_tied_prune;
    h->tune_chunk         = t_chunk;
    h->tune_fused_waves   = t_fused_waves;
    h->tune_fr            = t_fr;
    h->tune_simd_mfma     = t_simd_mfma;
    h->tune_dist_list     = t_dist_list;
    h->tune_near_fused    = t_near_fused;
    h->tune_fused_pack    = t_fused_pack;
    h->tune_screen_kernel = t_screen_kernel;
    h->use_graphs         = t_graph;
    h->contract_fma       = t_contract == "fma";
    h->ctx     = ctx;
    h->dim     = m->dim;
    h->n_mix   = m->n_mix;
    h->n_dens  = m->n_dens;
    h->n_mean  = m->n_mean;
    h->n_cov   = m->n_cov;
    h->nk      = nk;
    h->mix_off.assign(m->mix_offsets, m->mix_offsets + m->n_mix + 1);
    h->h_k_dens.assign(m->dens_index, m->dens_index + nk);
    h->h_d_mean.assign(m->dens_mean, m->dens_mean + m->n_dens);
    h->h_d_cov.assign(m->dens_cov, m->dens_cov + m->n_dens);

    // ---- model preparation (Mm/MixtureFeatureScorerElement.cc:21-33,
    //      Mm/CovarianceFeatureScorerElement.cc:21-51, Mm/Utilities.hh:53-91)
    h->m2lw.resize(nk);
    for (size_t k = 0; k < nk; ++k) {
        float minus2 = (float)(-2 * m->log_weight[k]);  // f64 product stored as Score
        h->m2lw[k]   = minus2 * (float)m->mixture_weight_scale;  // mixtureWeightScale_ is a Score
    }
    const float gs = (float)std::sqrt(m->gaussian_scale);  // gaussianScale_(std::sqrt(param)): f64 parameter, f64 root, f32 member
    h->isr.resize((size_t)m->n_cov * m->dim);
    h->lognorm.resize(m->n_cov);
    for (int c = 0; c < m->n_cov; ++c) {
        const float* var  = m->variances + (size_t)c * m->dim;
        double       lsum = 0;
        for (int i = 0; i < m->dim; ++i) {
            float inv                      = (float)1 / (float)std::sqrt((double)var[i]);
            h->isr[(size_t)c * m->dim + i] = inv * gs;
            lsum += std::log((double)std::fabs(var[i]));
        }
        // gaussLogNormFactor (Mm/Utilities.hh:70-75): N * log(2 pi) + logNorm; one vfmadd in the reference's default build
        float ln      = (float)(h->contract_fma ? std::fma((double)m->dim, std::log((double)2 * M_PI), lsum) : (double)m->dim * std::log((double)2 * M_PI) + lsum);
        h->lognorm[c] = ln * (gs * gs);
    }
    std::vector<uint32_t> k_mean(nk), k_cov(nk), k_dens(nk);
    std::vector<double>   c64(nk);
    std::vector<float>    c32(nk);
    for (size_t k = 0; k < nk; ++k) {
        uint32_t d = m->dens_index[k];
        k_dens[k]  = d;
        k_mean[k]  = m->dens_mean[d];
        k_cov[k]   = m->dens_cov[d];
        c64[k]     = (double)h->m2lw[k] + (double)h->lognorm[k_cov[k]];
        c32[k]     = h->m2lw[k] + h->lognorm[k_cov[k]];
    }
    // tied model: each density is referenced by several mixtures -> compute distances once
    h->tied = nk >= (size_t)4 * (size_t)m->n_dens;
    if (h->tied) {
        const uint32_t K = m->mix_offsets[1] - m->mix_offsets[0];
        bool           u = K > 0 && nk == (size_t)K * (size_t)m->n_mix;
        for (int i = 0; u && i < m->n_mix; ++i)
            u = (m->mix_offsets[i + 1] - m->mix_offsets[i] == K) &&
                (i == 0 || memcmp(m->dens_index + m->mix_offsets[i], m->dens_index, (size_t)K * 4) == 0);
        h->uniform = u;
        h->K       = (int)K;
    }

    int r = AMX_OK;
    if (!ctx) {
        *out = h;
        return AMX_OK;
    }
    hipSetDevice(ctx->device);
    if ((r = gupload(&h->d_mix_off, h->mix_off.data(), h->mix_off.size())) != AMX_OK ||
        (r = gupload(&h->d_k_mean, k_mean.data(), nk)) != AMX_OK || (r = gupload(&h->d_k_cov, k_cov.data(), nk)) != AMX_OK ||
        (r = gupload(&h->d_k_dens, k_dens.data(), nk)) != AMX_OK ||
        (r = gupload(&h->d_d_mean, m->dens_mean, (size_t)m->n_dens)) != AMX_OK ||
        (r = gupload(&h->d_d_cov, m->dens_cov, (size_t)m->n_dens)) != AMX_OK ||
        (r = gupload(&h->d_k_c64, c64.data(), nk)) != AMX_OK || (r = gupload(&h->d_k_c32, c32.data(), nk)) != AMX_OK ||
        (r = gupload(&h->d_means, m->means, (size_t)m->n_mean * m->dim)) != AMX_OK ||
        (r = gupload(&h->d_isr, h->isr.data(), h->isr.size())) != AMX_OK) {
        amx_gmm_destroy(h);
        return r;
    }
    h->pooled = (m->n_cov == 1);
    if (h->pooled) {
        // BatchFloatFeatureScorer::init: unscaled covariance element, c = logNormFactor - 2 * logWeight
        std::vector<float> isr0(m->dim), smeans((size_t)m->n_mean * m->dim), kc(nk);
        double             lsum = 0;
        for (int i = 0; i < m->dim; ++i) {
            isr0[i] = (float)1 / (float)std::sqrt((double)m->variances[i]);
            lsum += std::log((double)std::fabs(m->variances[i]));
        }
        const float ln = (float)(h->contract_fma ? std::fma((double)m->dim, std::log((double)2 * M_PI), lsum) : (double)m->dim * std::log((double)2 * M_PI) + lsum);
        for (int j = 0; j < m->n_mean; ++j)
            for (int i = 0; i < m->dim; ++i)
                smeans[(size_t)j * m->dim + i] = m->means[(size_t)j * m->dim + i] * isr0[i];
        for (size_t k = 0; k < nk; ++k)
            kc[k] = (float)((double)ln - 2 * m->log_weight[k]);
        h->h_smeans = smeans;  // host copies for the density clustering of preselection-batch-float
        h->h_k_mean = k_mean;
        if ((r = gupload(&h->d_isr0, isr0.data(), isr0.size())) != AMX_OK || (r = gupload(&h->d_smeans, smeans.data(), smeans.size())) != AMX_OK ||
            (r = gupload(&h->d_k_const, kc.data(), kc.size())) != AMX_OK) {
            amx_gmm_destroy(h);
            return r;
        }
    }
    if (h->uniform) {
        h->mix_pad = (h->n_mix + 63) & ~63;
        std::vector<float>  wt((size_t)h->K * h->mix_pad, 0.f);
        std::vector<double> ln64(h->K);
        std::vector<float>  ln32(h->K);
        for (int i = 0; i < h->n_mix; ++i)
            for (int k = 0; k < h->K; ++k)
                wt[(size_t)k * h->mix_pad + i] = h->m2lw[(size_t)i * h->K + k];
        for (int k = 0; k < h->K; ++k) {
            ln32[k] = h->lognorm[k_cov[k]];
            ln64[k] = (double)ln32[k];
        }
        {  // density -> list position, when that is a function (gmm_dist_kernel then writes the pruned scorer's frame-major image itself)
            std::vector<uint32_t> pos((size_t)h->n_dens, 0xffffffffu);
            bool                  once = true;
            for (int k = 0; k < h->K && once; ++k) {
                once = k_dens[k] < (uint32_t)h->n_dens && pos[k_dens[k]] == 0xffffffffu;
                if (once)
                    pos[k_dens[k]] = (uint32_t)k;
            }
            if (once && (r = gupload(&h->d_dens_pos, pos.data(), pos.size())) != AMX_OK) {
                amx_gmm_destroy(h);
                return r;
            }
            if (once) {  // the list's means and inverse deviations, transposed: lane = list position reads them coalesced
                const int          Kpad = (h->K + 63) & ~63, dim = m->dim;
                std::vector<float> mt((size_t)dim * Kpad, 0.f), it((size_t)dim * Kpad, 0.f);
                for (int k = 0; k < h->K; ++k)
                    for (int i = 0; i < dim; ++i) {
                        mt[(size_t)i * Kpad + k] = m->means[(size_t)k_mean[k] * dim + i];
                        it[(size_t)i * Kpad + k] = h->isr[(size_t)k_cov[k] * dim + i];
                    }
                if ((r = gupload(&h->d_means_t, mt.data(), mt.size())) != AMX_OK || (r = gupload(&h->d_isr_t, it.data(), it.size())) != AMX_OK) {
                    amx_gmm_destroy(h);
                    return r;
                }
            }
        }
        // screen tables (gmm_tied_tile_kernel): an f32 image of the per-entry constant and its largest magnitude per mixture
        std::vector<float> ahat((size_t)h->K * h->mix_pad, 0.f), amax(h->mix_pad, 0.f);
        for (int k = 0; k < h->K; ++k)
            for (int i = 0; i < h->n_mix; ++i) {
                const float a                    = wt[(size_t)k * h->mix_pad + i] + ln32[k];
                ahat[(size_t)k * h->mix_pad + i] = a;
                amax[i]                          = std::max(amax[i], std::fabs(a));
            }
        if ((r = gupload(&h->d_m2lw_t, wt.data(), wt.size())) != AMX_OK || (r = gupload(&h->d_ln64, ln64.data(), ln64.size())) != AMX_OK ||
            (r = gupload(&h->d_ln32, ln32.data(), ln32.size())) != AMX_OK || (r = gupload(&h->d_ahat_t, ahat.data(), ahat.size())) != AMX_OK ||
            (r = gupload(&h->d_amax, amax.data(), amax.size())) != AMX_OK ||
            (r = amx_internal_gmm_tied_create(h->K, h->n_mix, h->mix_pad, ahat.data(), &h->d_amin, &h->d_aup)) != AMX_OK) {
            amx_gmm_destroy(h);
            return r;
        }
        if (hipMalloc((void**)&h->d_tied_surv, 257 * 8) != hipSuccess || hipMemset(h->d_tied_surv, 0, 257 * 8) != hipSuccess ||
            hipHostMalloc((void**)&h->h_tied_surv, 257 * 8) != hipSuccess) {
            amx::set_error("amx_gmm_create: out of memory (tied-model statistics)");
            amx_gmm_destroy(h);
            return AMX_ERR_DEVICE;
        }
        memset(h->h_tied_surv, 0, 257 * 8);
    }
    // ---- MFMA screen tables (gmm_screen_kernel): private densities, <= 16 per mixture, operand fits f16
    if (!h->tied && screen_dim_supported(m->dim) && h->tune_screen) {
        uint32_t kmax = 0;
        for (int i = 0; i < m->n_mix; ++i)
            kmax = std::max(kmax, m->mix_offsets[i + 1] - m->mix_offsets[i]);
        const int d = m->dim, Kd = h->pooled ? d : 2 * d;
        if (kmax >= 1 && kmax <= 16 && Kd + 2 <= 128) {  // two more K columns carry the per-density constant as c_hi + c_lo
            const int Kp = Kd + 2 <= 64 ? 64 : 128, Rpad = (m->n_mix * 16 + 255) / 256 * 256, Mp = Rpad / 16;
            std::vector<_Float16> A((size_t)Rpad * Kp, (_Float16)0.f);
            std::vector<size_t>   row2(Kp == 64 ? (size_t)Rpad : 0);  // row of the old order -> row of gmm_screen_rows_kernel's order
            std::vector<float>    c((size_t)Rpad, std::numeric_limits<float>::infinity()), na(Mp, 0.f), cabs(Mp, 0.f), ra(Mp, 0.f);
            double                na_all = 0;
            bool                  fits = true;
            double                rmax2 = 0;
            for (size_t i = 0; i < h->isr.size(); ++i)
                rmax2 = std::max(rmax2, (double)h->isr[i] * (double)h->isr[i]);
            for (int i = 0; i < m->n_mix && fits; ++i)
                for (uint32_t k = m->mix_offsets[i]; k < m->mix_offsets[i + 1]; ++k) {
                    const uint32_t slot = k - m->mix_offsets[i];  // row inside the mixture's 16: see gmm_screen_epilogue
                    const size_t   row  = (size_t)i * 16 + (((slot >> 2) & 1) * 8 + (slot >> 3) * 4 + (slot & 3));
                    if (Kp == 64)
                        row2[row] = (size_t)(i >> 4) * 256 + (size_t)((i & 15) >> 1) * 32 + (slot >> 2) * 8 + (i & 1) * 4 + (slot & 3) + 1;
                    const float* mu  = m->means + (size_t)k_mean[k] * d;
                    const float* is  = h->isr.data() + (size_t)k_cov[k] * d;
                    double       cc = c64[k], n2 = 0, res2 = 0, true2 = 0;  // squared norms: rounded row, its f16 residual, the exact row
                    for (int x = 0; x < d; ++x) {
                        const double mr = (double)mu[x] * (double)is[x];
                        cc += mr * mr;
                        double a0, a1 = 0;
                        if (h->pooled)
                            a0 = -2.0 * mr;
                        else {
                            a0 = -2.0 * mr * (double)is[x];
                            a1 = (double)is[x] * (double)is[x];
                        }
                        if (!(std::fabs(a0) <= 65504.0 && a1 <= 65504.0))
                            fits = false;
                        const _Float16 h0 = (_Float16)(float)a0, h1 = (_Float16)(float)a1;
                        A[row * Kp + x]   = h0;
                        n2 += (double)(float)h0 * (double)(float)h0;
                        res2 += (a0 - (double)(float)h0) * (a0 - (double)(float)h0);
                        true2 += a0 * a0;
                        if (!h->pooled) {
                            A[row * Kp + d + x] = h1;
                            n2 += (double)(float)h1 * (double)(float)h1;
                            res2 += (a1 - (double)(float)h1) * (a1 - (double)(float)h1);
                            true2 += a1 * a1;
                        }
                    }
                    c[row]  = (float)cc;
                    na[i]   = std::max(na[i], (float)(std::sqrt(n2) * 1.0000001));
                    ra[i]   = std::max(ra[i], (float)(std::sqrt(res2) * 1.000001));
                    na_all  = std::max(na_all, std::sqrt(true2) * 1.000001);
                    cabs[i] = std::max(cabs[i], std::fabs((float)cc));
                    if (!std::isfinite(cc) || std::fabs(cc) > 65000.0)
                        fits = false;
                    const _Float16 chi = (_Float16)(float)cc;
                    A[row * Kp + Kd]     = chi;
                    A[row * Kp + Kd + 1] = (_Float16)(float)(cc - (double)(float)chi);
                }
            for (size_t row = 0; row < (size_t)Rpad; ++row)  // empty slots: +inf (never a survivor unless the frame keeps all)
                if (!std::isfinite(c[row]))
                    A[row * Kp + Kd] = (_Float16)std::numeric_limits<float>::infinity();
            const float sK = std::sqrt((float)Kd);
            for (int i = 0; i < Mp; ++i) {  // na -> p1, cabs -> p2 (see gmm_screen_epilogue)
                const float a = na[i], cb = cabs[i];
                na[i]   = 2.05f * ra[i] + 1.3e-4f * sK;
                cabs[i] = 1.3e-4f * sK * a + 1.6e-5f * cb;
            }
            if (fits) {
                h->scr_Kp = Kp;
                h->scr_Rpad = Rpad;
                h->scr_Mpad16 = Mp;
                h->scr_rmax2 = (float)(rmax2 * 1.000001);
                h->scr_na_all = (float)(2.05 * na_all * 1.000001);
                if (Kp == 64) {  // the same rows in the second kernel's order; rows that hold no density keep the +inf constant
                    std::vector<_Float16> A2((size_t)Rpad * Kp, (_Float16)0.f);
                    for (size_t row = 0; row < (size_t)Rpad; ++row)
                        A2[row * Kp + Kd] = (_Float16)std::numeric_limits<float>::infinity();
                    for (size_t row = 0; row < (size_t)Rpad; ++row)
                        if (row2[row])
                            memcpy(&A2[(row2[row] - 1) * Kp], &A[row * Kp], (size_t)Kp * sizeof(_Float16));
                    if ((r = gupload(&h->d_scr_A2, A2.data(), A2.size())) != AMX_OK) {
                        amx_gmm_destroy(h);
                        return r;
                    }
                    if (amx_internal_gmm_fused_supported(d, h->pooled ? 1 : 0, Kp) &&
                        (r = amx_internal_gmm_fused_create(d, m->n_mix, Rpad / 256, A2.data(), m->mix_offsets, k_mean.data(), c64.data(), m->means,
                                                           na.data(), cabs.data(), &h->d_fus_rec, &h->fus_rec_bytes)) != AMX_OK) {
                        amx_gmm_destroy(h);
                        return r;
                    }
                    if (h->d_fus_rec) {
                        const std::vector<unsigned long long> zero(256, 0ull);
                        if ((r = gupload(&h->d_fus_surv, zero.data(), zero.size())) != AMX_OK) {
                            amx_gmm_destroy(h);
                            return r;
                        }
                    }
                }
                if ((r = gupload(&h->d_scr_A, A.data(), A.size())) != AMX_OK || (r = gupload(&h->d_scr_c, c.data(), c.size())) != AMX_OK ||
                    (r = gupload(&h->d_scr_na, na.data(), na.size())) != AMX_OK || (r = gupload(&h->d_scr_cabs, cabs.data(), cabs.size())) != AMX_OK) {
                    amx_gmm_destroy(h);
                    return r;
                }
                h->screen = true;
            }
        }
    }
    // ---- SIMD-diagonal-maximum / batch-int tables (quantised means, integer constants; gmm_simd.hip) are built on the first
    // call of those scorers (ensure_simd): most handles never use them, and a model they cannot represent must not keep the
    // float scorers from being created.  Host copies of the model for that build:
    h->h_means.assign(m->means, m->means + (size_t)m->n_mean * m->dim);
    h->h_vars.assign(m->variances, m->variances + (size_t)m->n_cov * m->dim);
    h->h_logw.assign(m->log_weight, m->log_weight + nk);
    h->mws = m->mixture_weight_scale;
    h->gsc = m->gaussian_scale;
    *out = h;
    return AMX_OK;
}

void amx_gmm_destroy(amx_gmm* h) {
    if (!h)
        return;
    if (!h->ctx) {
        delete h;
        return;
    }
    hipSetDevice(h->ctx->device);
    for (auto& kv : h->graphs)
        if (kv.second)
            hipGraphExecDestroy(kv.second);
    amx_internal_gmm_simd_destroy(h->simd);
    amx_internal_gmm_presel_destroy(h->presel);
    hipFree(h->d_mix_off);
    hipFree(h->d_k_mean);
    hipFree(h->d_k_cov);
    hipFree(h->d_k_dens);
    hipFree(h->d_dens_pos);
    hipFree(h->d_means_t);
    hipFree(h->d_isr_t);
    hipFree(h->d_d_mean);
    hipFree(h->d_d_cov);
    hipFree(h->d_k_c64);
    hipFree(h->d_k_c32);
    hipFree(h->d_means);
    hipFree(h->d_isr);
    hipFree(h->d_smeans);
    hipFree(h->d_k_const);
    hipFree(h->d_isr0);
    hipFree(h->d_dist);
    hipFree(h->d_dist64);
    hipFree(h->d_scr_A);
    hipFree(h->d_scr_A2);
    hipFree(h->d_fus_rec);
    hipFree(h->d_fus_surv);
    hipFree(h->d_best32);
    hipFree(h->d_scr_c);
    hipFree(h->d_scr_na);
    hipFree(h->d_scr_cabs);
    hipFree(h->d_scr_X);
    hipFree(h->d_scr_nx);
    hipFree(h->d_scr_q);
    hipFree(h->d_scr_masks);
    hipFree(h->d_scr_pmin);
    hipFree(h->d_scr_pidx);
    hipFree(h->d_host_f);
    hipFree(h->d_host_s);
    hipFree(h->d_host_b);
    hipFree(h->d_m2lw_t);
    hipFree(h->d_ahat_t);
    hipFree(h->d_amax);
    hipFree(h->d_amin);
    hipFree(h->d_aup);
    hipFree(h->d_tied_ws);
    hipFree(h->d_tied_surv);
    if (h->h_tied_surv)
        hipHostFree(h->h_tied_surv);
    hipFree(h->d_ln64);
    hipFree(h->d_ln32);
    delete h;
}

int amx_gmm_n_mixtures(const amx_gmm* h) {
    return h ? h->n_mix : 0;
}
int amx_gmm_dimension(const amx_gmm* h) {
    return h ? h->dim : 0;
}

int amx_gmm_tables(const amx_gmm* h, float* m2lw, float* isr, float* lognorm) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gmm_tables: NULL handle");
    if (m2lw)
        memcpy(m2lw, h->m2lw.data(), h->m2lw.size() * 4);
    if (isr)
        memcpy(isr, h->isr.data(), h->isr.size() * 4);
    if (lognorm)
        memcpy(lognorm, h->lognorm.data(), h->lognorm.size() * 4);
    return AMX_OK;
}

extern "C++" {
static int ensure_simd(amx_gmm* h) {
    if (h->simd_status == 1)
        return AMX_OK;
    if (h->simd_status < 0) {
        amx::set_error("SIMD-diagonal-maximum tables of this model could not be built");
        return h->simd_status;
    }
    amx_gmm_model v;
    memset(&v, 0, sizeof v);
    v.dim = h->dim;
    v.n_mix = h->n_mix;
    v.n_dens = h->n_dens;
    v.n_mean = h->n_mean;
    v.n_cov = h->n_cov;
    v.mix_offsets = h->mix_off.data();
    v.dens_index = h->h_k_dens.data();
    v.log_weight = h->h_logw.data();
    v.dens_mean = h->h_d_mean.data();
    v.dens_cov = h->h_d_cov.data();
    v.means = h->h_means.data();
    v.variances = h->h_vars.data();
    v.mixture_weight_scale = h->mws;
    v.gaussian_scale = h->gsc;
    v.tuning = h->tune_simd_mfma ? nullptr : "simd_mfma=0";
    const int r = amx_internal_gmm_simd_create(&v, h->contract_fma ? 1 : 0, &h->simd, nullptr);
    h->simd_status = r == AMX_OK ? 1 : r;
    return r;
}
}  // extern "C++"

// dense or pruned for this call of a shared-list tied model: amx_gmm_model.tuning tied_prune=0 forces the dense kernel, =1 the pruned path,
// default adaptive -- the pruned kernel counts the (density, frame, tile) triples it had to evaluate, the host reads the count of
// EARLIER calls from pinned memory (no synchronisation) and stays on the dense kernel for 64 calls while more than 10 % stood
// The 2 KB copy costs 4 us of stream time, a thirtieth of a decoder-sized pruned pass, and the decision below looks at windows of
// many calls: every 8th pruned call publishes the counters (issued after the graph replay, never recorded in it).
static int tied_publish(amx_gmm* h) {
    if ((++h->tied_copy_tick & 7u) != 0)
        return AMX_OK;
    AMX_HIP(hipMemcpyAsync(h->h_tied_surv, h->d_tied_surv, 257 * 8, hipMemcpyDeviceToHost, h->ctx->stream));
    return AMX_OK;
}

static bool tied_decide_prune(amx_gmm* h) {
    const int forced = h->tune_tied_prune;
    if (forced >= 0)
        return forced != 0;
    // one asynchronous copy delivers survivors and examined triples of the calls that have COMPLETED: a consistent pair, however far
    // the host runs ahead
    unsigned long long seen = 0;
    for (int i = 0; i < 256; ++i)
        seen += ((volatile unsigned long long*)h->h_tied_surv)[i];
    const unsigned long long examined = ((volatile unsigned long long*)h->h_tied_surv)[256];
    if (h->tied_dense_calls > 0) {
        --h->tied_dense_calls;
        return false;
    }
    if (examined - h->tied_triples >= (1ull << 20) && seen >= h->tied_seen) {
        const double frac = (double)(seen - h->tied_seen) / (double)(examined - h->tied_triples);
        h->tied_seen      = seen;
        h->tied_triples   = examined;
        if (frac > 0.10) {
            h->tied_dense_calls = 64;
            return false;
        }
    }
    return true;
}

int amx_gmm_score_dev(amx_gmm* h, int mode, const float* feats_dev, int T, float* scores_dev, uint32_t* best_dev) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gmm_score_dev: NULL handle");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_gmm_score_dev: host-only handle (created without a context)");
    AMX_REQUIRE(mode == AMX_GMM_MAX || mode == AMX_GMM_SUM || mode == AMX_GMM_BATCH_FLOAT || mode == AMX_GMM_SIMD || mode == AMX_GMM_BATCH_INT ||
                        mode == AMX_GMM_PRESELECTION_FLOAT || mode == AMX_GMM_PRESELECTION_INT,
                AMX_ERR_INVALID,
                "amx_gmm_score_dev: unknown mode %d", mode);
    AMX_REQUIRE(T >= 0, AMX_ERR_INVALID, "amx_gmm_score_dev: negative frame count");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_dev && scores_dev, AMX_ERR_INVALID, "amx_gmm_score_dev: NULL buffer");
    AMX_HIP(hipSetDevice(h->ctx->device));
    const int fblocks = amx::ceil_div(T, 256);
    // contract=fma covers every mode (round 6): the float scorers fuse the distance's accumulate, the preselection scorer also its
    // clustering distances (Mm::unrolledVectorDistance: one vfmadd231ss per term in the default build); the quantised scorers' arithmetic
    // is integer -- their one f64 site, gaussLogNormFactor's N * log(2 pi) + logNorm, follows the contract on the host (gmm_simd.hip)
    if (mode == AMX_GMM_SIMD || mode == AMX_GMM_BATCH_INT || mode == AMX_GMM_PRESELECTION_INT) {
        const int r = ensure_simd(h);
        if (r != AMX_OK)
            return r;
    }
    if (mode == AMX_GMM_PRESELECTION_INT) {
        AMX_REQUIRE(best_dev == nullptr, AMX_ERR_UNSUPPORTED, "amx_gmm_score_dev: preselection-batch-int does not assign densities");
        const int r = amx_internal_gmm_simd_presel_build(h->simd, h->ctx, h->presel_clusters, h->presel_select, h->presel_iterations);
        if (r != AMX_OK)
            return r;
        return amx_internal_gmm_simd_presel_score(h->simd, h->ctx, feats_dev, T, scores_dev);
    }
    if (mode == AMX_GMM_SIMD)
        return amx_internal_gmm_simd_score(h->simd, h->ctx, 0, feats_dev, T, scores_dev, best_dev);
    if (mode == AMX_GMM_BATCH_INT) {
        AMX_REQUIRE(best_dev == nullptr, AMX_ERR_UNSUPPORTED, "amx_gmm_score_dev: batch-diagonal-maximum-int does not assign densities");
        return amx_internal_gmm_simd_score(h->simd, h->ctx, 1, feats_dev, T, scores_dev, nullptr);
    }
    if (mode == AMX_GMM_PRESELECTION_FLOAT) {
        AMX_REQUIRE(h->pooled, AMX_ERR_INVALID, "amx_gmm_score_dev: feature scorer supports only globally pooled covariance");
        AMX_REQUIRE(best_dev == nullptr, AMX_ERR_UNSUPPORTED, "amx_gmm_score_dev: preselection-batch-float does not assign densities");
        if (!h->presel) {
            const int r = amx_internal_gmm_presel_create(h->ctx, h->dim, h->nk, h->h_k_mean.data(), h->h_smeans.data(), h->d_smeans, h->d_k_mean,
                                                         h->presel_clusters, h->presel_select, h->presel_iterations, h->presel_backoff, h->contract_fma ? 1 : 0, &h->presel);
            if (r != AMX_OK)
                return r;
        }
        return amx_internal_gmm_presel_score(h->presel, h->ctx, feats_dev, T, scores_dev, h->d_mix_off, h->d_k_mean, h->d_k_const, h->d_smeans,
                                             h->d_isr0, h->n_mix);
    }
    if (mode == AMX_GMM_BATCH_FLOAT) {
        // Mm::BatchFloatFeatureScorer::init: criticalError("feature scorer supports only globally pooled covariance")
        AMX_REQUIRE(h->pooled, AMX_ERR_INVALID, "amx_gmm_score_dev: feature scorer supports only globally pooled covariance");
        AMX_REQUIRE(best_dev == nullptr, AMX_ERR_UNSUPPORTED, "amx_gmm_score_dev: batch-diagonal-maximum-float does not assign densities");
        int mt = 16;
        while (mt > 4 && (long)amx::ceil_div(h->n_mix, mt) * fblocks < 2048)
            mt /= 2;
        amx::GmmDims           dims{T, h->dim, h->n_mix, mt};
        dim3                   grid(amx::ceil_div(h->n_mix, mt), fblocks);
        amx::ScopedKernelTimer timer(h->ctx, "gmm");
        switch (h->dim) {
#define AMX_GMM_CASE(D)                                                                                                   \
    case D:                                                                                                               \
        hipLaunchKernelGGL((h->contract_fma ? amx::gmm_batch_float_kernel<D, true> : amx::gmm_batch_float_kernel<D, false>), grid, dim3(256), 0, h->ctx->stream, feats_dev, scores_dev,    \
                           h->d_mix_off, h->d_k_mean, h->d_k_const, h->d_smeans, h->d_isr0, dims);                         \
        break;
            AMX_GMM_CASE(16)
            AMX_GMM_CASE(24)
            AMX_GMM_CASE(32)
            AMX_GMM_CASE(33)
            AMX_GMM_CASE(39)
            AMX_GMM_CASE(40)
            AMX_GMM_CASE(45)
            AMX_GMM_CASE(48)
            AMX_GMM_CASE(64)
#undef AMX_GMM_CASE
            default:  // any other dimension: the same arithmetic with the feature row re-read from memory
                hipLaunchKernelGGL((h->contract_fma ? amx::gmm_batch_float_kernel<0, true> : amx::gmm_batch_float_kernel<0, false>), grid, dim3(256), 0, h->ctx->stream, feats_dev, scores_dev, h->d_mix_off,
                                   h->d_k_mean, h->d_k_const, h->d_smeans, h->d_isr0, dims);
                break;
        }
        AMX_HIP(hipGetLastError());
        return AMX_OK;
    }
    if (!h->tied && h->screen && mode == AMX_GMM_MAX) {
        // three launches and their gaps are a fifth of a 256-frame pass: replay them as a graph (not while profiling: the
        // per-launch events are not part of the graph).  First call plain (sizes the workspaces), second call captures.
        // (a pass that IS one launch -- the fused kernel packing its own operand rows, one chunk -- is cheaper launched than replayed:
        // 0.0316 against 0.0369 ms per 256 frames, round 6)
        const bool one_launch = h->d_fus_rec && h->tune_fused && !h->tune_screen_all && h->tune_fused_pack && h->tune_fused_waves != 13 &&
                                T <= h->tune_chunk;
        if (!(h->use_graphs && !h->ctx->profiling && T <= 4096) || one_launch)
            return score_screened(h, feats_dev, T, scores_dev, best_dev, false, nullptr, nullptr, nullptr);
        const amx_gmm::GraphKey key{feats_dev, scores_dev, best_dev, h->ctx->stream, T};
        auto                    it = h->graphs.find(key);
        if (it == h->graphs.end()) {
            if (h->graphs.size() >= 64) {  // a caller that never repeats a signature: stop caching instead of growing the map
                for (auto& kv : h->graphs)
                    if (kv.second)
                        hipGraphExecDestroy(kv.second);
                h->graphs.clear();
                h->use_graphs = 0;
            }
            else
                h->graphs[key] = nullptr;
            return score_screened(h, feats_dev, T, scores_dev, best_dev, false, nullptr, nullptr, nullptr);
        }
        if (it->second == nullptr) {
            hipGraph_t g = nullptr;
            if (h->graphs.size() > 64 || hipStreamBeginCapture(h->ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
                (void)hipGetLastError();
                h->use_graphs = 0;  // a caller that keeps changing buffers, or a stream that cannot capture
                return score_screened(h, feats_dev, T, scores_dev, best_dev, false, nullptr, nullptr, nullptr);
            }
            const int      r  = score_screened(h, feats_dev, T, scores_dev, best_dev, false, nullptr, nullptr, nullptr);
            const bool     ok = hipStreamEndCapture(h->ctx->stream, &g) == hipSuccess && r == AMX_OK && g != nullptr;
            hipGraphExec_t ex = nullptr;
            if (!ok || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGetLastError();
                if (g)
                    hipGraphDestroy(g);
                h->use_graphs = 0;
                return score_screened(h, feats_dev, T, scores_dev, best_dev, false, nullptr, nullptr, nullptr);
            }
            hipGraphDestroy(g);
            it->second = ex;
        }
        else if (h->count_survivors && h->d_fus_rec)  // a replay runs the counting kernel without passing score_screened's bookkeeping
            h->fus_pairs += (unsigned long long)T * (unsigned long long)h->n_mix;
        AMX_HIP(hipGraphLaunch(it->second, h->ctx->stream));
        return AMX_OK;
    }
    if (!h->tied) {
        amx::GmmParams p;
        p.feats   = feats_dev;
        p.scores  = scores_dev;
        p.best    = best_dev;
        p.mix_off = h->d_mix_off;
        p.k_mean  = h->d_k_mean;
        p.k_cov   = h->d_k_cov;
        p.k_c64   = h->d_k_c64;
        p.k_c32   = h->d_k_c32;
        p.means   = h->d_means;
        p.isr     = h->d_isr;
        p.T       = T;
        p.dim     = h->dim;
        p.n_mix   = h->n_mix;
        // enough workgroups to fill 256 CUs several times over, but >= 64 B of scores per row
        int mt = 16;
        while (mt > 4 && (long)amx::ceil_div(h->n_mix, mt) * fblocks < 2048)
            mt /= 2;
        p.mix_tile = mt;
        dim3                   grid(amx::ceil_div(h->n_mix, mt), fblocks);
        amx::ScopedKernelTimer timer(h->ctx, "gmm");
        return mode == AMX_GMM_MAX ? launch_direct<amx::MaxState>(h, p, grid) : launch_direct<amx::SumState>(h, p, grid);
    }
    // ---- tied: chunk frames so the distance scratch stays <= 256 MB
    const int chunk_max = (int)std::max<size_t>(256, std::min<size_t>(16384, ((size_t)64 << 20) / (size_t)h->n_dens / 256 * 256));
    // The pruned path of a shared-list model is six launches (and, every 8th call, a 2 KB copy): at the decoder's batch sizes their gaps are a seventh
    // of the pass, so repeated passes on unchanged buffers are replayed as one HIP graph like the screened CART path above (the
    // dense / pruned decision stays outside: a graph is only recorded and replayed for the pruned path).
    if (h->uniform && mode == AMX_GMM_MAX && h->tied_forced < 0 && T <= chunk_max && T <= 4096 && h->tune_screen) {
        const bool prune = tied_decide_prune(h);
        auto       nested = [&](int forced) {
            h->tied_forced = forced;
            const int r    = amx_gmm_score_dev(h, mode, feats_dev, T, scores_dev, best_dev);
            h->tied_forced = -1;
            return r;
        };
        if (!prune || !h->use_graphs || h->ctx->profiling)
            return nested(prune ? 1 : 0);
        const amx_gmm::GraphKey key{feats_dev, scores_dev, best_dev, h->ctx->stream, T};
        auto                    it = h->graphs.find(key);
        if (it == h->graphs.end()) {  // first pass with this signature: plain launches (they size the workspaces)
            if (h->graphs.size() >= 64) {
                for (auto& kv : h->graphs)
                    if (kv.second)
                        hipGraphExecDestroy(kv.second);
                h->graphs.clear();
                h->use_graphs = 0;
            }
            else
                h->graphs[key] = nullptr;
            return nested(1);
        }
        if (it->second == nullptr) {
            hipGraph_t gr = nullptr;
            if (hipStreamBeginCapture(h->ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
                (void)hipGetLastError();
                h->use_graphs = 0;
                return nested(1);
            }
            h->tied_capturing = true;
            const int r       = nested(1);
            h->tied_capturing = false;
            const bool     ok = hipStreamEndCapture(h->ctx->stream, &gr) == hipSuccess && r == AMX_OK && gr != nullptr;
            hipGraphExec_t ex = nullptr;
            if (!ok || hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGetLastError();
                if (gr)
                    hipGraphDestroy(gr);
                h->use_graphs = 0;
                return nested(1);
            }
            hipGraphDestroy(gr);
            h->graphs[key] = ex;  // (by key: a nested call that had to grow a scratch buffer empties the map)
            it             = h->graphs.find(key);
        }
        else {  // replay: the statistics the nested call would have kept
            h->tied_rep_triples += (unsigned long long)h->K * (unsigned long long)T * (unsigned long long)(h->mix_pad / 64);
        }
        AMX_HIP(hipGraphLaunch(it->second, h->ctx->stream));
        return tied_publish(h);
    }
    for (int t0 = 0; t0 < T; t0 += chunk_max) {
        const int Tc   = std::min(chunk_max, T - t0);
        const int Tpad = (Tc + 63) & ~63;
        size_t    need = (size_t)h->n_dens * Tpad;
        if (need > h->dist_floats) {
            for (auto& kv : h->graphs)  // recorded passes hold the old scratch addresses
                if (kv.second)
                    hipGraphExecDestroy(kv.second);
            h->graphs.clear();
            hipFree(h->d_dist);
            h->d_dist      = nullptr;
            h->dist_floats = 0;
            AMX_HIP(hipMalloc((void**)&h->d_dist, need * sizeof(float)));
            h->dist_floats = need;
        }
        amx::GmmDistParams dp;
        dp.feats     = feats_dev + (size_t)t0 * h->dim;
        dp.dist      = h->d_dist;
        dp.d_mean    = h->d_d_mean;
        dp.d_cov     = h->d_d_cov;
        dp.means     = h->d_means;
        dp.isr       = h->d_isr;
        dp.T         = Tc;
        dp.Tpad      = Tpad;
        dp.dim       = h->dim;
        dp.n_dens    = h->n_dens;
        const int  fb      = amx::ceil_div(Tc, 256);
        // densities per workgroup: 16, fewer while that leaves the chip short of workgroups (a 256-frame batch of config 3 is ONE
        // frame block: 256 workgroups of 16 densities each wait for 16 scalar fetches in a row; 1024 of 4 do not)
        const bool stage   = (long)amx::ceil_div(h->n_dens, 16) * fb < 4L * std::max(h->ctx->n_cu, 1);
        dp.dens_tile       = stage ? (int)std::min<long>(16, std::max<long>(2, (long)h->n_dens * fb / (4L * std::max(h->ctx->n_cu, 1)))) : 16;
        const bool use_uni = h->uniform;
        const int screen = h->tune_screen;  // 0: plain f64 kernel (A/B runs, tests)
        const bool need64  = use_uni && mode == AMX_GMM_MAX && !screen;
        if (need64 && need > h->dist64_cap) {
            hipFree(h->d_dist64);
            h->d_dist64   = nullptr;
            h->dist64_cap = 0;
            AMX_HIP(hipMalloc((void**)&h->d_dist64, need * sizeof(double)));
            h->dist64_cap = need;
        }
        // pruned exact path of a shared-list model (gmm_tied.hip) unless the survivor statistics of earlier calls say that this model /
        // these features leave too much standing (tied_decide_prune); decided here because the distance kernel then also writes the
        // frame-major image that path works on
        const bool prune = use_uni && mode == AMX_GMM_MAX && screen && (h->tied_forced >= 0 ? h->tied_forced == 1 : tied_decide_prune(h));
        float*              dt   = nullptr;
        unsigned long long* near = nullptr;
        if (prune) {
            const size_t need_ws = amx_internal_gmm_tied_workspace(h->K, Tc, h->mix_pad);
            if (need_ws > h->tied_ws_cap) {
                for (auto& kv : h->graphs)
                    if (kv.second)
                        hipGraphExecDestroy(kv.second);
                h->graphs.clear();
                hipFree(h->d_tied_ws);
                h->d_tied_ws   = nullptr;
                h->tied_ws_cap = 0;
                AMX_HIP(hipMalloc(&h->d_tied_ws, need_ws));
                h->tied_ws_cap     = need_ws;
                h->tied_keys_clean = false;
            }
            dt = amx_internal_gmm_tied_dt(h->d_tied_ws, h->K, Tc, h->d_dens_pos != nullptr);
        }
        {
            amx::ScopedKernelTimer timer(h->ctx, "gmm_dist");
            const bool list_order = dt && !need64 && h->d_means_t && h->tune_dist_list;
            if (list_order && h->tune_near_fused && !(h->tied_capturing && !h->tied_keys_clean)) {
                near = amx_internal_gmm_tied_near(h->d_tied_ws);
                if (near && !h->tied_keys_clean) {  // a new workspace, or a call that did not get as far as putting the keys back
                    int ri = amx_internal_gmm_tied_near_init(h->ctx, h->d_tied_ws);
                    if (ri != AMX_OK)
                        return ri;
                }
                h->tied_keys_clean = false;  // until amx_internal_gmm_tied_score has been enqueued behind this kernel
            }
            int r = list_order ? launch_dist_list(h, feats_dev + (size_t)t0 * h->dim, Tc, dt, near) : AMX_ERR_STATE;
            if (r == AMX_ERR_STATE) {
                if (near)
                    h->tied_keys_clean = true;  // nothing touched them
                near = nullptr;
            }
            if (r == AMX_ERR_STATE)  // (a dimension without an instance, or not the pruned one-pass path)
                r = launch_dist(h, dp, dim3(amx::ceil_div(h->n_dens, dp.dens_tile), fb), need64 ? h->d_dist64 : nullptr, stage, dt,
                                (h->K + 63) & ~63);
            if (r != AMX_OK)
                return r;
        }
        if (use_uni) {
            int FR = h->tune_fr;
            amx::GmmUniformDims    ud{Tc, Tpad, h->n_mix, h->mix_pad, h->K, 0};
            dim3                   grid(amx::ceil_div(h->n_mix, 256), amx::ceil_div(Tc, FR));
            float*                 sc = scores_dev + (size_t)t0 * h->n_mix;
            uint32_t*              bd = best_dev ? best_dev + (size_t)t0 * h->n_mix : nullptr;
            amx::ScopedKernelTimer timer(h->ctx, "gmm_combine");
#define AMX_UNI(STATE, F)                                                                                                        \
    hipLaunchKernelGGL((amx::gmm_combine_uniform_kernel<amx::STATE, F>), grid, dim3(256), 0, h->ctx->stream, h->d_dist, h->d_dist64, \
                       sc, bd, h->d_m2lw_t, h->d_k_dens, h->d_ln64, h->d_ln32, ud)
            if (mode == AMX_GMM_MAX && screen) {
                if (prune) {
                    int r = amx_internal_gmm_tied_score(h->ctx, h->d_dist, h->d_k_dens, h->K, Tc, Tpad, h->n_mix, h->mix_pad, h->d_aup,
                                                        h->d_amax, h->d_m2lw_t, h->d_ahat_t, h->d_ln64, h->d_ln32, h->d_amin, h->d_tied_ws, sc, bd,
                                                        h->d_tied_surv, dt != nullptr, near != nullptr);
                    if (r != AMX_OK)
                        return r;
                    if (near)
                        h->tied_keys_clean = true;  // tied_list_kernel puts them back
                    if (!h->tied_capturing && (r = tied_publish(h)) != AMX_OK)
                        return r;
                    h->tied_rep_triples += (unsigned long long)h->K * (unsigned long long)Tc * (unsigned long long)(h->mix_pad / 64);
                }
                else
                    hipLaunchKernelGGL(amx::gmm_tied_tile_kernel, dim3(h->mix_pad / 64, Tpad / 64), dim3(256), 0, h->ctx->stream, h->d_dist, sc,
                                       bd, h->d_m2lw_t, h->d_ahat_t, h->d_amax, h->d_k_dens, h->d_ln64, ud);
            }
            else if (mode == AMX_GMM_MAX) {
                if (FR == 4) AMX_UNI(MaxState, 4);
                else if (FR == 16) AMX_UNI(MaxState, 16);
                else if (FR == 2) AMX_UNI(MaxState, 2);
                else AMX_UNI(MaxState, 8);
            }
            else {
                FR = 8;
                grid = dim3(amx::ceil_div(h->n_mix, 256), amx::ceil_div(Tc, FR));
                AMX_UNI(SumState, 8);
            }
#undef AMX_UNI
            AMX_HIP(hipGetLastError());
            continue;
        }
        amx::GmmCombineParams cp;
        cp.dist     = h->d_dist;
        cp.scores   = scores_dev + (size_t)t0 * h->n_mix;
        cp.best     = best_dev ? best_dev + (size_t)t0 * h->n_mix : nullptr;
        cp.mix_off  = h->d_mix_off;
        cp.k_dens   = h->d_k_dens;
        cp.k_c64    = h->d_k_c64;
        cp.k_c32    = h->d_k_c32;
        cp.T        = Tc;
        cp.Tpad     = Tpad;
        cp.n_mix    = h->n_mix;
        cp.mix_tile = 4;
        dim3                   grid(amx::ceil_div(h->n_mix, cp.mix_tile), fb);
        amx::ScopedKernelTimer timer(h->ctx, "gmm_combine");
        amx::GmmCombineDims    cd{cp.T, cp.Tpad, cp.n_mix, cp.mix_tile};
        if (mode == AMX_GMM_MAX)
            hipLaunchKernelGGL((amx::gmm_combine_kernel<amx::MaxState>), grid, dim3(256), 0, h->ctx->stream, cp.dist, cp.scores,
                               cp.best, cp.mix_off, cp.k_dens, cp.k_c64, cp.k_c32, cd);
        else
            hipLaunchKernelGGL((amx::gmm_combine_kernel<amx::SumState>), grid, dim3(256), 0, h->ctx->stream, cp.dist, cp.scores,
                               cp.best, cp.mix_off, cp.k_dens, cp.k_c64, cp.k_c32, cd);
        AMX_HIP(hipGetLastError());
    }
    return AMX_OK;
}

extern "C" int amx_stats_accumulate_dev(amx_ctx*, const float*, int, int, uint32_t*, unsigned long long*, double*);

int amx_gmm_score_stats_dev(amx_gmm* h, const float* feats_dev, int T, float* scores_dev, uint32_t* best_density_dev, uint32_t* best_state_dev,
                            unsigned long long* state_counts_dev, double* score_sum_dev) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gmm_score_stats_dev: NULL handle");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_gmm_score_stats_dev: host-only handle (created without a context)");
    AMX_REQUIRE(state_counts_dev && score_sum_dev, AMX_ERR_INVALID, "amx_gmm_score_stats_dev: NULL accumulator");
    AMX_REQUIRE(T >= 0, AMX_ERR_INVALID, "amx_gmm_score_stats_dev: negative frame count");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_dev && scores_dev, AMX_ERR_INVALID, "amx_gmm_score_stats_dev: NULL buffer");
    AMX_HIP(hipSetDevice(h->ctx->device));
    if (!h->tied && h->screen)  // arg-min over the states fused into the exact stage: the score matrix is not read again
        return score_screened(h, feats_dev, T, scores_dev, best_density_dev, true, best_state_dev, state_counts_dev, score_sum_dev);
    int r = amx_gmm_score_dev(h, AMX_GMM_MAX, feats_dev, T, scores_dev, best_density_dev);
    if (r != AMX_OK)
        return r;
    return amx_stats_accumulate_dev(h->ctx, scores_dev, T, h->n_mix, best_state_dev, state_counts_dev, score_sum_dev);
}

// byte form of the best-density matrix: a quarter of the u32 form's memory, the same time (tools/gmm_store_ab.py: the kernel pays
// for tracking the best density, not for its bytes).  The fused kernel writes it directly; every other path scores into a u32
// workspace and narrows it.
int amx_gmm_score_stats_u8_dev(amx_gmm* h, const float* feats_dev, int T, float* scores_dev, uint8_t* best_density_dev, uint32_t* best_state_dev,
                               unsigned long long* state_counts_dev, double* score_sum_dev) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gmm_score_stats_u8_dev: NULL handle");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_gmm_score_stats_u8_dev: host-only handle (created without a context)");
    AMX_REQUIRE(state_counts_dev && score_sum_dev, AMX_ERR_INVALID, "amx_gmm_score_stats_u8_dev: NULL accumulator");
    AMX_REQUIRE(T >= 0, AMX_ERR_INVALID, "amx_gmm_score_stats_u8_dev: negative frame count");
    for (size_t m = 0; m + 1 < h->mix_off.size(); ++m)
        AMX_REQUIRE(h->mix_off[m + 1] - h->mix_off[m] <= 255u, AMX_ERR_UNSUPPORTED,
                    "amx_gmm_score_stats_u8_dev: mixture %zu has %u densities (a byte holds an index below 255)", m, h->mix_off[m + 1] - h->mix_off[m]);
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_dev && scores_dev && best_density_dev, AMX_ERR_INVALID, "amx_gmm_score_stats_u8_dev: NULL buffer");
    AMX_HIP(hipSetDevice(h->ctx->device));
    const bool fused = !h->tied && h->screen && h->d_fus_rec && h->tune_fused && !h->tune_screen_all;
    bool       direct = fused;
    for (int t0 = 0; t0 < T && direct; t0 += h->tune_chunk) {  // every chunk of the pass on an 8- or 12-wave kernel
        const int nw = amx_internal_gmm_fused_waves((std::min(h->tune_chunk, T - t0) + 255) / 256 * 256, h->tune_fused_waves);
        direct       = nw == 8 || nw == 12;
    }
    if (direct)
        return score_screened(h, feats_dev, T, scores_dev, (uint32_t*)best_density_dev, true, best_state_dev, state_counts_dev, score_sum_dev, 1);
    const size_t need = (size_t)T * h->n_mix;
    if (need > h->best32_cap) {
        hipFree(h->d_best32);
        h->d_best32   = nullptr;
        h->best32_cap = 0;
        AMX_HIP(hipMalloc((void**)&h->d_best32, need * 4));
        h->best32_cap = need;
    }
    int r = amx_gmm_score_stats_dev(h, feats_dev, T, scores_dev, h->d_best32, best_state_dev, state_counts_dev, score_sum_dev);
    if (r != AMX_OK)
        return r;
    hipLaunchKernelGGL(amx::best_narrow_kernel, dim3((unsigned)std::min<size_t>((need + 1023) / 1024, 65536)), dim3(256), 0, h->ctx->stream,
                       h->d_best32, best_density_dev, need);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_gmm_screen_counts(amx_gmm* h, int enable, unsigned long long* survivors, unsigned long long* pairs) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gmm_screen_counts: NULL handle");
    if (survivors)
        *survivors = 0;
    if (pairs)
        *pairs = 0;
    if (h->d_tied_surv) {  // tied model on the pruned path: (density, frame, 64-mixture tile) triples that survived / were submitted
        AMX_HIP(hipSetDevice(h->ctx->device));
        AMX_HIP(hipStreamSynchronize(h->ctx->stream));
        unsigned long long c[256], v = 0;
        AMX_HIP(hipMemcpy(c, h->d_tied_surv, sizeof c, hipMemcpyDeviceToHost));
        for (int i = 0; i < 256; ++i)
            v += c[i];
        if (survivors)
            *survivors = v - h->tied_rep_seen;
        if (pairs)
            *pairs = h->tied_rep_triples;
        h->tied_rep_seen    = v;
        h->tied_rep_triples = 0;
        return AMX_OK;
    }
    if (!h->d_fus_surv)  // not the fused screened path: nothing is counted
        return AMX_OK;
    AMX_HIP(hipSetDevice(h->ctx->device));
    AMX_HIP(hipStreamSynchronize(h->ctx->stream));
    unsigned long long v = 0, part[256];
    AMX_HIP(hipMemcpy(part, h->d_fus_surv, sizeof part, hipMemcpyDeviceToHost));
    for (int i = 0; i < 256; ++i)
        v += part[i];
    if (survivors)
        *survivors = v;
    if (pairs)
        *pairs = h->fus_pairs;
    AMX_HIP(hipMemset(h->d_fus_surv, 0, 256 * 8));
    h->fus_pairs = 0;
    if (h->count_survivors != (enable != 0)) {  // captured passes carry the counter argument they were recorded with
        for (auto& kv : h->graphs)
            if (kv.second)
                hipGraphExecDestroy(kv.second);
        h->graphs.clear();
    }
    h->count_survivors = enable != 0;
    return AMX_OK;
}

int amx_gmm_set_preselection(amx_gmm* h, int clusters, int select_clusters, int iterations, float backoff_score) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gmm_set_preselection: NULL handle");
    AMX_REQUIRE(clusters >= 1 && clusters <= 256 && select_clusters >= 1 && select_clusters <= 256 && iterations >= 0, AMX_ERR_INVALID,
                "amx_gmm_set_preselection: clusters and select-clusters must be in 1..256");
    amx_internal_gmm_presel_destroy(h->presel);  // rebuilt with the new parameters on the next preselection call
    h->presel            = nullptr;
    h->presel_clusters   = clusters;
    h->presel_select     = select_clusters;
    h->presel_iterations = iterations;
    h->presel_backoff    = backoff_score;
    return AMX_OK;
}

int amx_gmm_preselection_clustering(amx_gmm* h, int* n_clusters, uint32_t* cluster_of, float* cluster_means) {
    AMX_REQUIRE(h && h->ctx, AMX_ERR_INVALID, "amx_gmm_preselection_clustering: NULL / host-only handle");
    AMX_REQUIRE(h->pooled, AMX_ERR_INVALID, "amx_gmm_preselection_clustering: feature scorer supports only globally pooled covariance");
    if (!h->presel) {
        AMX_HIP(hipSetDevice(h->ctx->device));
        const int r = amx_internal_gmm_presel_create(h->ctx, h->dim, h->nk, h->h_k_mean.data(), h->h_smeans.data(), h->d_smeans, h->d_k_mean,
                                                     h->presel_clusters, h->presel_select, h->presel_iterations, h->presel_backoff, h->contract_fma ? 1 : 0, &h->presel);
        if (r != AMX_OK)
            return r;
    }
    return amx_internal_gmm_presel_info(h->presel, n_clusters, cluster_of, cluster_means);
}

int amx_gmm_preselection_int_clustering(amx_gmm* h, int* n_clusters, uint32_t* cluster_of, float* cluster_means) {
    AMX_REQUIRE(h && h->ctx, AMX_ERR_INVALID, "amx_gmm_preselection_int_clustering: NULL / host-only handle");
    int r = ensure_simd(h);
    if (r != AMX_OK)
        return r;
    if ((r = amx_internal_gmm_simd_presel_build(h->simd, h->ctx, h->presel_clusters, h->presel_select, h->presel_iterations)) != AMX_OK)
        return r;
    return amx_internal_gmm_simd_presel_info(h->simd, n_clusters, cluster_of, cluster_means);
}

float amx_gmm_simd_scaling(const amx_gmm* h) {
    if (!h || !h->ctx || ensure_simd(const_cast<amx_gmm*>(h)) != AMX_OK)
        return 0.f;
    return amx_internal_gmm_simd_scaling(h->simd);
}

long amx_gmm_accumulator_size(const amx_gmm* h) {
    return h ? (long)h->nk + (long)h->n_mean * (1 + h->dim) + (long)h->n_cov * (1 + h->dim) : 0;
}

// ---- "MIXSET" accumulator files (Mm::MixtureSetEstimator::write / read, binary, version 2):
//   char[8] "MIXSET\0\0" | u32 version | u32 dimension
//   u32 nMeans       { u32 dim, f64 sum[dim], f64 weight }          Mm/VectorAccumulator.hh:80-100
//   u32 nCovariances { u32 dim, f64 sumOfSquares[dim], f64 weight }
//   u32 nDensities   { u32 meanIndex, u32 covarianceIndex }          Mm/GaussDensityEstimator.cc:46-62
//   u32 nMixtures    { u32 n, { u32 densityIndex, f64 weight } x n } Mm/MixtureEstimator.cc:140-170
// (Mm/AbstractMixtureSetEstimator.cc:404-508; little endian like every Core::BinaryStream file).  The flat accumulator keeps
// the model's own index order, which is what the reference's index maps produce for an estimator built from that model.
extern "C++" {
namespace {
struct AccLayout {
    long long off_mw, off_ms, off_cw, off_cs;
};
AccLayout acc_layout(const amx_gmm* h) {
    AccLayout l;
    l.off_mw = (long long)h->nk;
    l.off_ms = l.off_mw + h->n_mean;
    l.off_cw = l.off_ms + (long long)h->n_mean * h->dim;
    l.off_cs = l.off_cw + h->n_cov;
    return l;
}
template<class T>
bool put(FILE* f, T v) {
    return fwrite(&v, sizeof(T), 1, f) == 1;  // little-endian host
}
template<class T>
bool get(FILE* f, T* v) {
    return fread(v, sizeof(T), 1, f) == 1;
}
}  // namespace
}  // extern "C++"

int amx_gmm_accumulator_write(const amx_gmm* h, const double* acc, const char* path) {
    AMX_REQUIRE(h && acc && path, AMX_ERR_INVALID, "amx_gmm_accumulator_write: NULL argument");
    FILE* f = fopen(path, "wb");
    AMX_REQUIRE(f, AMX_ERR_INVALID, "amx_gmm_accumulator_write: cannot open '%s'", path);
    const AccLayout l = acc_layout(h);
    const char      magic[8] = {'M', 'I', 'X', 'S', 'E', 'T', 0, 0};
    bool            ok = fwrite(magic, 1, 8, f) == 8 && put(f, (uint32_t)2) && put(f, (uint32_t)h->dim);
    ok                 = ok && put(f, (uint32_t)h->n_mean);
    for (int i = 0; ok && i < h->n_mean; ++i)
        ok = put(f, (uint32_t)h->dim) && fwrite(acc + l.off_ms + (long long)i * h->dim, 8, (size_t)h->dim, f) == (size_t)h->dim &&
             put(f, acc[l.off_mw + i]);
    ok = ok && put(f, (uint32_t)h->n_cov);
    for (int i = 0; ok && i < h->n_cov; ++i)
        ok = put(f, (uint32_t)h->dim) && fwrite(acc + l.off_cs + (long long)i * h->dim, 8, (size_t)h->dim, f) == (size_t)h->dim &&
             put(f, acc[l.off_cw + i]);
    ok = ok && put(f, (uint32_t)h->n_dens);
    for (int d = 0; ok && d < h->n_dens; ++d)
        ok = put(f, h->h_d_mean[d]) && put(f, h->h_d_cov[d]);
    ok = ok && put(f, (uint32_t)h->n_mix);
    for (int m = 0; ok && m < h->n_mix; ++m) {
        ok = put(f, (uint32_t)(h->mix_off[m + 1] - h->mix_off[m]));
        for (uint32_t k = h->mix_off[m]; ok && k < h->mix_off[m + 1]; ++k)
            ok = put(f, h->h_k_dens[k]) && put(f, acc[k]);
    }
    ok = (fclose(f) == 0) && ok;
    AMX_REQUIRE(ok, AMX_ERR_INVALID, "amx_gmm_accumulator_write: write to '%s' failed", path);
    return AMX_OK;
}

int amx_gmm_accumulator_read(const amx_gmm* h, const char* path, double* acc) {
    AMX_REQUIRE(h && acc && path, AMX_ERR_INVALID, "amx_gmm_accumulator_read: NULL argument");
    FILE* f = fopen(path, "rb");
    AMX_REQUIRE(f, AMX_ERR_INVALID, "amx_gmm_accumulator_read: cannot open '%s'", path);
    const AccLayout l = acc_layout(h);
    char            magic[8] = {0};
    uint32_t        version = 0, dim = 0, n = 0;
    const char*     why = nullptr;
    bool            ok = fread(magic, 1, 8, f) == 8 && get(f, &version) && get(f, &dim);
    if (ok && strncmp(magic, "MIXSET", 7) != 0)
        why = "not a MIXSET estimator file";  // the reference: 'Mixture set estimator file with magic "..." could not be read'
    else if (ok && version == 0)
        why = "version 0 files (integer counts) are not supported";
    else if (ok && (int)dim != h->dim)
        why = "dimension differs from the model";
    auto vec = [&](double* sums, double* weight) {
        uint32_t size = 0;
        return get(f, &size) && (int)size == h->dim && fread(sums, 8, (size_t)h->dim, f) == (size_t)h->dim && get(f, weight);
    };
    ok = ok && !why && get(f, &n) && (int)n == h->n_mean;
    for (int i = 0; ok && i < h->n_mean; ++i)
        ok = vec(acc + l.off_ms + (long long)i * h->dim, acc + l.off_mw + i);
    ok = ok && get(f, &n) && (int)n == h->n_cov;
    for (int i = 0; ok && i < h->n_cov; ++i)
        ok = vec(acc + l.off_cs + (long long)i * h->dim, acc + l.off_cw + i);
    ok = ok && get(f, &n) && (int)n == h->n_dens;
    for (int d = 0; ok && d < h->n_dens; ++d) {
        uint32_t mi = 0, ci = 0;
        ok          = get(f, &mi) && get(f, &ci) && mi == h->h_d_mean[d] && ci == h->h_d_cov[d];
    }
    ok = ok && get(f, &n) && (int)n == h->n_mix;
    for (int m = 0; ok && m < h->n_mix; ++m) {
        ok = get(f, &n) && n == h->mix_off[m + 1] - h->mix_off[m];
        for (uint32_t k = h->mix_off[m]; ok && k < h->mix_off[m + 1]; ++k) {
            uint32_t d = 0;
            ok         = get(f, &d) && d == h->h_k_dens[k] && get(f, &acc[k]);
        }
    }
    fclose(f);
    if (why) {
        amx::set_error("amx_gmm_accumulator_read: '%s': %s", path, why);
        return AMX_ERR_INVALID;
    }
    AMX_REQUIRE(ok, AMX_ERR_INVALID, "amx_gmm_accumulator_read: '%s' is truncated or its topology differs from the model", path);
    return AMX_OK;
}

int amx_gmm_best_density_dev(amx_gmm* h, const float* feats_dev, int T, const uint32_t* mixture_dev, uint32_t* best_density_dev,
                             float* scores_dev) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gmm_best_density_dev: NULL handle");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_gmm_best_density_dev: host-only handle (created without a context)");
    AMX_REQUIRE(T >= 0, AMX_ERR_INVALID, "amx_gmm_best_density_dev: negative frame count");
    AMX_REQUIRE(h->d_k_mean && h->d_k_cov && h->d_k_c64 && h->d_means && h->d_isr, AMX_ERR_UNSUPPORTED,
                "amx_gmm_best_density_dev: the model's scorer keeps no diagonal-maximum tables on the device");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_dev && mixture_dev && best_density_dev, AMX_ERR_INVALID, "amx_gmm_best_density_dev: NULL buffer");
    AMX_HIP(hipSetDevice(h->ctx->device));
    amx::ScopedKernelTimer timer(h->ctx, "gmm_best_density");
    hipLaunchKernelGGL((h->contract_fma ? amx::gmm_best_density_kernel<true> : amx::gmm_best_density_kernel<false>), dim3((T + 255) / 256), dim3(256), 0, h->ctx->stream, feats_dev, mixture_dev, best_density_dev,
                       scores_dev, h->d_mix_off, h->d_k_mean, h->d_k_cov, h->d_k_c64, h->d_means, h->d_isr, T, h->dim, h->n_mix);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_gmm_accumulate_dev(amx_gmm* h, const float* feats_dev, int T, const uint32_t* mixture_dev, const uint32_t* best_density_dev,
                           int best_density_ld, double* acc_dev) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gmm_accumulate_dev: NULL handle");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_gmm_accumulate_dev: host-only handle (created without a context)");
    AMX_REQUIRE(T >= 0 && (best_density_ld == 0 || best_density_ld >= h->n_mix), AMX_ERR_INVALID, "amx_gmm_accumulate_dev: bad shape");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_dev && mixture_dev && best_density_dev && acc_dev, AMX_ERR_INVALID, "amx_gmm_accumulate_dev: NULL buffer");
    AMX_HIP(hipSetDevice(h->ctx->device));
    const long long off_mw = (long long)h->nk, off_ms = off_mw + h->n_mean, off_cw = off_ms + (long long)h->n_mean * h->dim,
                    off_cs = off_cw + h->n_cov;
    const int              pooled = (h->n_cov == 1 && h->dim <= 256) ? 1 : 0;
    const int              blocks = (T + 255) / 256;
    amx::ScopedKernelTimer timer(h->ctx, "gmm_accumulate");
    hipLaunchKernelGGL(amx::gmm_accumulate_kernel, dim3(blocks), dim3(256), 0, h->ctx->stream, feats_dev, mixture_dev, best_density_dev,
                       (const unsigned char*)nullptr, best_density_ld, T, h->dim, h->n_mix, h->d_mix_off, h->d_k_dens, h->d_d_mean, h->d_d_cov, acc_dev,
                       off_mw, off_ms, off_cw, off_cs, pooled);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_gmm_accumulate_u8_dev(amx_gmm* h, const float* feats_dev, int T, const uint32_t* mixture_dev, const uint8_t* best_density_dev,
                              int best_density_ld, double* acc_dev) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gmm_accumulate_u8_dev: NULL handle");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_gmm_accumulate_u8_dev: host-only handle (created without a context)");
    AMX_REQUIRE(T >= 0 && (best_density_ld == 0 || best_density_ld >= h->n_mix), AMX_ERR_INVALID, "amx_gmm_accumulate_u8_dev: bad shape");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_dev && mixture_dev && best_density_dev && acc_dev, AMX_ERR_INVALID, "amx_gmm_accumulate_u8_dev: NULL buffer");
    AMX_HIP(hipSetDevice(h->ctx->device));
    const long long off_mw = (long long)h->nk, off_ms = off_mw + h->n_mean, off_cw = off_ms + (long long)h->n_mean * h->dim,
                    off_cs = off_cw + h->n_cov;
    const int              pooled = (h->n_cov == 1 && h->dim <= 256) ? 1 : 0;
    const int              blocks = (T + 255) / 256;
    amx::ScopedKernelTimer timer(h->ctx, "gmm_accumulate");
    hipLaunchKernelGGL(amx::gmm_accumulate_kernel, dim3(blocks), dim3(256), 0, h->ctx->stream, feats_dev, mixture_dev, (const uint32_t*)nullptr,
                       (const unsigned char*)best_density_dev, best_density_ld, T, h->dim, h->n_mix, h->d_mix_off, h->d_k_dens, h->d_d_mean, h->d_d_cov,
                       acc_dev, off_mw, off_ms, off_cw, off_cs, pooled);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_gmm_accumulate_weighted_dev(amx_gmm* h, int mode, const float* feats_dev, int T, const uint32_t* mixture_dev,
                                    const double* weight_dev, const uint32_t* best_density_dev, int best_density_ld, double* acc_dev) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gmm_accumulate_weighted_dev: NULL handle");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_gmm_accumulate_weighted_dev: host-only handle (created without a context)");
    AMX_REQUIRE(mode == AMX_GMM_VITERBI || mode == AMX_GMM_BAUM_WELCH, AMX_ERR_INVALID, "amx_gmm_accumulate_weighted_dev: unknown mode %d", mode);
    AMX_REQUIRE(T >= 0 && (best_density_ld == 0 || best_density_ld >= h->n_mix), AMX_ERR_INVALID, "amx_gmm_accumulate_weighted_dev: bad shape");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_dev && mixture_dev && acc_dev && (mode == AMX_GMM_BAUM_WELCH || best_density_dev), AMX_ERR_INVALID,
                "amx_gmm_accumulate_weighted_dev: NULL buffer");
    if (mode == AMX_GMM_BAUM_WELCH) {
        uint32_t kmax = 0;
        for (int m = 0; m < h->n_mix; ++m)
            kmax = std::max(kmax, h->mix_off[m + 1] - h->mix_off[m]);
        AMX_REQUIRE(kmax <= (uint32_t)amx::kBwMaxDens, AMX_ERR_UNSUPPORTED,
                    "amx_gmm_accumulate_weighted_dev: Baum-Welch statistics support up to %d densities per mixture (model has %u)",
                    amx::kBwMaxDens, kmax);
    }
    AMX_HIP(hipSetDevice(h->ctx->device));
    const long long off_mw = (long long)h->nk, off_ms = off_mw + h->n_mean, off_cw = off_ms + (long long)h->n_mean * h->dim,
                    off_cs = off_cw + h->n_cov;
    const int              pooled = (h->n_cov == 1) ? 1 : 0;
    const int              per_block = 4 * amx::kBwFramesPerWave;
    const int              blocks = (T + per_block - 1) / per_block;
    amx::ScopedKernelTimer timer(h->ctx, "gmm_accumulate_weighted");
    hipLaunchKernelGGL((h->contract_fma ? amx::gmm_accumulate_weighted_kernel<true> : amx::gmm_accumulate_weighted_kernel<false>), dim3(blocks), dim3(256), 0, h->ctx->stream, mode, feats_dev, mixture_dev,
                       weight_dev, best_density_dev, best_density_ld, T, h->dim, h->n_mix, h->d_mix_off, h->d_k_dens, h->d_d_mean, h->d_d_cov,
                       h->d_k_c32, h->d_means, h->d_isr, acc_dev, off_mw, off_ms, off_cw, off_cs, pooled);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_gmm_score(amx_gmm* h, int mode, const float* feats_host, int T, float* scores_host, uint32_t* best_host) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_gmm_score: NULL handle");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_gmm_score: host-only handle (created without a context)");
    AMX_REQUIRE(T >= 0, AMX_ERR_INVALID, "amx_gmm_score: negative frame count");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_host && scores_host, AMX_ERR_INVALID, "amx_gmm_score: NULL buffer");
    AMX_HIP(hipSetDevice(h->ctx->device));
    // staging buffers live in the handle and only grow: the decoder calls this once per ring-buffer fill, and a
    // hipMalloc / hipFree pair per call costs more than scoring a small batch
    hipStream_t  st = h->ctx->stream;
    const size_t nf = (size_t)T * h->dim, ns = (size_t)T * h->n_mix;
    auto grow = [h](void** p, size_t* cap, size_t need) {
        if (need <= *cap)
            return true;
        for (auto& kv : h->graphs)  // captured passes may hold the old staging addresses
            if (kv.second)
                hipGraphExecDestroy(kv.second);
        h->graphs.clear();
        hipFree(*p);
        *p   = nullptr;
        *cap = 0;
        if (hipMalloc(p, need) != hipSuccess)
            return false;
        *cap = need;
        return true;
    };
    if (!grow((void**)&h->d_host_f, &h->host_f_cap, nf * 4) || !grow((void**)&h->d_host_s, &h->host_s_cap, ns * 4) ||
        (best_host && !grow((void**)&h->d_host_b, &h->host_b_cap, ns * 4))) {
        (void)hipGetLastError();
        amx::set_error("amx_gmm_score: out of device memory");
        return AMX_ERR_DEVICE;
    }
    float*    d_f = h->d_host_f;
    float*    d_s = h->d_host_s;
    uint32_t* d_b = best_host ? h->d_host_b : nullptr;
    if (hipMemcpyAsync(d_f, feats_host, nf * 4, hipMemcpyHostToDevice, st) != hipSuccess) {
        amx::set_error("amx_gmm_score: H2D copy failed");
        return AMX_ERR_DEVICE;
    }
    int r = amx_gmm_score_dev(h, mode, d_f, T, d_s, d_b);
    if (r != AMX_OK)
        return r;
    if (hipMemcpyAsync(scores_host, d_s, ns * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        (best_host && hipMemcpyAsync(best_host, d_b, ns * 4, hipMemcpyDeviceToHost, st) != hipSuccess) ||
        hipStreamSynchronize(st) != hipSuccess) {
        amx::set_error("amx_gmm_score: D2H copy / kernel execution failed: %s", hipGetErrorString(hipGetLastError()));
        return AMX_ERR_DEVICE;
    }
    return AMX_OK;
}

}  // extern "C"
