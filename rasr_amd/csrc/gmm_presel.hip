// gmm_presel.hip -- Mm::BatchPreselectionFloatFeatureScorer ("preselection-batch-float", Mm/BatchFeatureScorer.cc:256-318) with
// Mm::FloatDensityClustering (Mm/DensityClustering.hh, .tcc): the batch-float scorer (pooled covariance, pre-scaled means)
// restricted, per frame, to the densities of the `select-clusters` clusters closest to the scaled feature; a mixture without an
// active density scores `backoff-score`.
//
//   build      k-means over the pre-scaled density means (one "density" per mixture entry, as in the reference's batch scorers):
//              initial clusters = densities drawn with srand(1) / rand() % nDensities (glibc's additive-feedback generator,
//              restated below: the library does not touch the process-wide rand() state), `iterations` rounds of
//                assign   every density to the first closest cluster (sequential f32 sum of squared differences, strict '<')
//                         -- a HIP kernel, lane = density, cluster means through the scalar cache
//                update   f64 component sums in density order / count -> f32, on the host (model-sized, like the reference's)
//   per call   cluster_select_kernel: lane = frame; distances to all clusters into LDS-free global scratch, then `select`
//              rounds of "smallest (distance, cluster) pair above the previous one" give the threshold pair; every cluster at or
//              below it is active (std::sort order; equal distances are broken by cluster index, which the reference leaves
//              unspecified); the result leaves transposed: one 64-bit lane mask per (wavefront of 64 frames, cluster)
//              presel_score_kernel: gmm_batch_float_kernel's arithmetic with the wave's lane mask of the density's cluster read
//              through the scalar cache -- a density whose cluster no frame of the wave selected is skipped altogether.
#include "common.hpp"

#include <algorithm>
#include "gmm_device.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace amx {

constexpr int kPreselMaxClusters = 256;  // Mm/DensityClustering.cc:21-22: parameter range 1..256 (ClusterIndex = u8)

// lane = density (mixture entry): first closest cluster
template<int DIM, bool FMA>   // FMA: Mm::unrolledVectorDistance<f32, f32> of the reference's default build -- score += df * df is one vfmadd231ss per term (read off libref_native.so)
__global__ __launch_bounds__(256) void presel_assign_kernel(const float* __restrict__ g_smeans, const uint32_t* __restrict__ g_k_mean, int nk,
                                                           const float* __restrict__ g_cm, int n_clusters, uint32_t* __restrict__ g_cluster_of,
                                                           int dim_rt) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nk)
        return;
    const int    dim = DIM > 0 ? DIM : dim_rt;
    const float* row = g_smeans + (size_t)g_k_mean[k] * dim;
    float        mu[DIM > 0 ? DIM : 1];  // DIM == 0 (a dimension without a specialised kernel): the row is re-read from memory
#pragma unroll
    for (int i = 0; i < DIM; ++i)
        mu[i] = row[i];
    float    bd = FLT_MAX;
    uint32_t bc = 0;
    for (int c = 0; c < n_clusters; ++c) {
        const float* cm = g_cm + (size_t)c * dim;  // wave-uniform: scalar loads
        float        score = 0.f;
        if (DIM > 0) {
#pragma unroll
            for (int i = 0; i < DIM; ++i) {  // unrolledVectorDistance(meanForCluster, meanForDensity): sequential, a = cluster mean
                const float df = cm[i] - mu[i];
                score          = mad<FMA>(df, df, score);
            }
        }
        else
            for (int i = 0; i < dim; ++i) {
                const float df = cm[i] - row[i];
                score          = mad<FMA>(df, df, score);
            }
        if (score < bd) {
            bd = score;
            bc = (uint32_t)c;
        }
    }
    g_cluster_of[k] = bc;
}

// lane = frame: distances of the scaled feature to every cluster mean -> scratch [n_clusters x Tpad] (coalesced along frames),
// threshold pair by `n_select` selection rounds over that column, lane masks per (wave, cluster)
template<int DIM, bool FMA>
__global__ __launch_bounds__(64) void cluster_select_kernel(const float* __restrict__ g_feats, const float* __restrict__ g_isr0, int T, int Tpad,
                                                           const float* __restrict__ g_cm, int n_clusters, int n_select,
                                                           float* __restrict__ g_dist, unsigned long long* __restrict__ g_masks, int dim_rt) {
    const int  lane = threadIdx.x;
    const int  t    = blockIdx.x * 64 + lane;
    const bool live = t < T;
    const int  tt   = live ? t : T - 1;
    const int  dim  = DIM > 0 ? DIM : dim_rt;
    ScaledRow<DIM> x;
    x.load(g_feats + (size_t)tt * dim, g_isr0, dim);  // setFeature: f * variance_ (1 / sigma)
    for (int c = 0; c < n_clusters; ++c) {
        const float* cm = g_cm + (size_t)c * dim;
        float        score = 0.f;
        if (DIM > 0) {
#pragma unroll
            for (int i = 0; i < DIM; ++i) {  // unrolledVectorDistance(feature, meanForCluster)
                const float df = x(i) - cm[i];
                score          = mad<FMA>(df, df, score);
            }
        }
        else
            for (int i = 0; i < dim; ++i) {
                const float df = x(i) - cm[i];
                score          = mad<FMA>(df, df, score);
            }
        g_dist[(size_t)c * Tpad + t] = score;
    }
    // the n_select-th smallest (distance, cluster) pair of this frame
    float pd = -1.f;  // distances are >= 0 (or NaN, which never compares below anything: such clusters stay inactive)
    int   pc = -1;
    for (int s = 0; s < n_select; ++s) {
        float bd = __builtin_inff();
        int   bc = n_clusters;
        bool  any = false;
        for (int c = 0; c < n_clusters; ++c) {
            const float d     = g_dist[(size_t)c * Tpad + t];
            const bool  above = d > pd || (d == pd && c > pc);
            const bool  lower = d < bd || (d == bd && c < bc);
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
    // A frame whose distances are all NaN (a NaN feature) ranks no cluster above another: the reference's std::sort leaves such a
    // list in index order and still marks the first n_select clusters (Mm/BatchFeatureScorer.cc selectClusters); so does this.
    const bool none = pc < 0;
    for (int c = 0; c < n_clusters; ++c) {
        const float d      = g_dist[(size_t)c * Tpad + t];
        const bool  active = live && (none ? c < n_select : (d < pd || (d == pd && c <= pc)));
        const unsigned long long m = __ballot(active);
        if (lane == 0)
            g_masks[(size_t)blockIdx.x * n_clusters + c] = m;
    }
}

template<int DIM, bool FMA>
__global__ __launch_bounds__(256) void presel_score_kernel(const float* __restrict__ g_feats, float* __restrict__ g_scores,
                                                          const uint32_t* __restrict__ g_mix_off, const uint32_t* __restrict__ g_k_mean,
                                                          const float* __restrict__ g_k_const, const float* __restrict__ g_smeans,
                                                          const float* __restrict__ g_isr0, const uint32_t* __restrict__ g_cluster_of,
                                                          const unsigned long long* __restrict__ g_masks, int n_clusters, float backoff, int T,
                                                          int n_mix, int mix_tile, int dim_rt) {
    const int  lane = threadIdx.x & 63;
    const int  wave = threadIdx.x >> 6;
    const int  wg   = blockIdx.y * 4 + wave;  // 64-frame group = row of the mask table
    const int  t    = wg * 64 + lane;
    if (wg * 64 >= T)
        return;
    const bool live = t < T;
    const int  tt   = live ? t : (T - 1);
    const int  dim  = DIM > 0 ? DIM : dim_rt;
    ScaledRow<DIM> x;
    x.load(g_feats + (size_t)tt * dim, g_isr0, dim);
    const unsigned long long* masks = g_masks + (size_t)wg * n_clusters;
    const int m0 = blockIdx.x * mix_tile;
    const int m1 = min(m0 + mix_tile, n_mix);
    for (int m = m0; m < m1; ++m) {
        const uint32_t k0 = g_mix_off[m], k1 = g_mix_off[m + 1];
        float          best = FLT_MAX;
        for (uint32_t k = k0; k < k1; ++k) {
            const unsigned long long am = masks[g_cluster_of[k]];  // wave-uniform
            if (am == 0ull)
                continue;  // no frame of this wave selected the density's cluster
            const float* mu = g_smeans + (size_t)g_k_mean[k] * dim;
            const float  r  = batch_float_distance<DIM, FMA>(mu, x, g_k_const[k], dim);
            const bool  act = (am >> lane) & 1ull;
            best            = act ? (best < r ? best : r) : best;  // _mm_min_ps(score, r), the reference's operand order: a NaN sum replaces the score
        }
        if (live)
            g_scores[(size_t)t * n_mix + m] = best < FLT_MAX ? 0.5f * best : (best == FLT_MAX ? backoff : best);  // (a NaN score is neither: it stays)
    }
}

// glibc's srand(seed) / rand() (TYPE_3 additive feedback generator, r[i] = r[i-3] + r[i-31]): what the reference's
// initializeClusters draws from.  Restated so that the library leaves the process-wide generator alone; the tests compare the
// clustering with a checker that calls libc's srand / rand.
struct GlibcRand {
    std::vector<int32_t> state;  // r[0 .. i-1]; output k is r[k + 344] >> 1
    explicit GlibcRand(unsigned seed) {
        int32_t r[31];
        r[0] = (int32_t)(seed ? seed : 1);
        for (int i = 1; i < 31; ++i) {
            const int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
            int64_t       w  = 16807 * lo - 2836 * hi;
            if (w < 0)
                w += 2147483647;
            r[i] = (int32_t)w;
        }
        state.assign(r, r + 31);
        for (int i = 31; i < 34; ++i)
            state.push_back(state[i - 31]);
        for (int i = 34; i < 344; ++i)
            state.push_back((int32_t)((uint32_t)state[i - 31] + (uint32_t)state[i - 3]));
    }
    int next() {
        const size_t   i = state.size();
        const uint32_t v = (uint32_t)state[i - 31] + (uint32_t)state[i - 3];
        state.push_back((int32_t)v);
        if (state.size() > 4096)  // only the last 31 values matter
            state.erase(state.begin(), state.end() - 64);
        return (int)(v >> 1);
    }
};

struct GmmPresel {
    bool      fma = false;   // contract=fma: clustering distances and the scorer's distance as the reference's default build fuses them
    int       dim = 0, n_clusters = 0, n_select = 0;
    size_t    nk = 0;
    float     backoff = 40000.f;
    float*    d_cm = nullptr;          // [n_clusters x dim]
    uint32_t* d_cluster_of = nullptr;  // [nk]
    float*    d_dist = nullptr;        // per-call scratch [n_clusters x Tpad]
    unsigned long long* d_masks = nullptr;
    int       cap_T = 0;
    std::vector<uint32_t> h_cluster_of;
    std::vector<float>    h_cm;
};

}  // namespace amx

extern "C" void amx_internal_gmm_presel_destroy(void* p) {
    amx::GmmPresel* s = (amx::GmmPresel*)p;
    if (!s)
        return;
    hipFree(s->d_cm);
    hipFree(s->d_cluster_of);
    hipFree(s->d_dist);
    hipFree(s->d_masks);
    delete s;
}

#define AMX_PRESEL_DIMS(X) X(16) X(24) X(32) X(33) X(39) X(40) X(45) X(48) X(64)

// smeans_host [n_mean x dim] pre-scaled means, k_mean_host [nk]; device copies of both are the scorer's own (d_smeans, d_k_mean)
extern "C" int amx_internal_gmm_presel_create(amx_ctx* ctx, int dim, size_t nk, const uint32_t* k_mean_host, const float* smeans_host,
                                              const float* d_smeans, const uint32_t* d_k_mean, int n_clusters, int n_select, int iterations,
                                              float backoff, int contract_fma, void** out) {
    using namespace amx;
    *out = nullptr;
    // DensityClusteringBase::init: "reducing number of clusters ... because there are too few densities"
    if ((size_t)n_clusters > nk)
        n_clusters = (int)nk;
    AMX_REQUIRE(n_clusters >= 1 && n_clusters <= kPreselMaxClusters, AMX_ERR_INVALID, "preselection: clusters must be in 1..256 (got %d)", n_clusters);
    AMX_REQUIRE(n_select >= 1 && n_select <= n_clusters, AMX_ERR_INVALID, "preselection: select-clusters (%d) must be in 1..clusters (%d)", n_select,
                n_clusters);
    AMX_REQUIRE(iterations >= 0, AMX_ERR_INVALID, "preselection: negative iteration count");
    GmmPresel* s = new GmmPresel;
    s->fma = contract_fma != 0;
    s->dim = dim;
    s->nk = nk;
    s->n_clusters = n_clusters;
    s->n_select = n_select;
    s->backoff = backoff;
    s->h_cm.assign((size_t)n_clusters * dim, 0.f);
    s->h_cluster_of.assign(nk, 0u);
    {  // initializeClusters
        GlibcRand         rng(1);
        std::vector<char> used(nk, 0);
        for (int c = 0; c < n_clusters; ++c) {
            uint32_t pick;
            do {
                pick = (uint32_t)rng.next() % (uint32_t)nk;
            } while (used[pick]);
            used[pick] = 1;
            memcpy(&s->h_cm[(size_t)c * dim], smeans_host + (size_t)k_mean_host[pick] * dim, (size_t)dim * 4);
        }
    }
    auto fail = [&](const char* what) {
        amx::set_error("preselection: %s failed: %s", what, hipGetErrorString(hipGetLastError()));
        amx_internal_gmm_presel_destroy(s);
        return AMX_ERR_DEVICE;
    };
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void**)&s->d_cm, s->h_cm.size() * 4) != hipSuccess ||
        hipMalloc((void**)&s->d_cluster_of, std::max<size_t>(nk, 1) * 4) != hipSuccess)
        return fail("device allocation");
    std::vector<double> sums((size_t)n_clusters * dim);
    std::vector<size_t> cnt(n_clusters);
    for (int it = 0; it < iterations; ++it) {
        if (hipMemcpyAsync(s->d_cm, s->h_cm.data(), s->h_cm.size() * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return fail("upload");
        const dim3 grid((unsigned)((nk + 255) / 256));
        switch (dim) {
#define X(D)                                                                                                                        \
    case D:                                                                                                                         \
        hipLaunchKernelGGL((s->fma ? presel_assign_kernel<D, true> : presel_assign_kernel<D, false>), grid, dim3(256), 0, ctx->stream, d_smeans, d_k_mean, (int)nk, s->d_cm, n_clusters, \
                           s->d_cluster_of, dim);                                                                                   \
        break;
            AMX_PRESEL_DIMS(X)
#undef X
            default:  // any other dimension: rows re-read from memory, the same sums
                hipLaunchKernelGGL((s->fma ? presel_assign_kernel<0, true> : presel_assign_kernel<0, false>), grid, dim3(256), 0, ctx->stream, d_smeans, d_k_mean, (int)nk, s->d_cm, n_clusters,
                                   s->d_cluster_of, dim);
                break;
        }
        if (hipMemcpyAsync(s->h_cluster_of.data(), s->d_cluster_of, nk * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            return fail("density assignment");
        // updateClusterMeans: f64 sums in density order
        std::fill(sums.begin(), sums.end(), 0.0);
        std::fill(cnt.begin(), cnt.end(), (size_t)0);
        for (size_t k = 0; k < nk; ++k) {
            const uint32_t c  = s->h_cluster_of[k];
            const float*   mu = smeans_host + (size_t)k_mean_host[k] * dim;
            double*        sm = &sums[(size_t)c * dim];
            for (int i = 0; i < dim; ++i)
                sm[i] = sm[i] + (double)mu[i];
            ++cnt[c];
        }
        for (int c = 0; c < n_clusters; ++c)
            if (cnt[c])
                for (int i = 0; i < dim; ++i)
                    s->h_cm[(size_t)c * dim + i] = (float)(sums[(size_t)c * dim + i] / (double)cnt[c]);
    }
    if (hipMemcpy(s->d_cm, s->h_cm.data(), s->h_cm.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(s->d_cluster_of, s->h_cluster_of.data(), nk * 4, hipMemcpyHostToDevice) != hipSuccess)
        return fail("upload");
    *out = s;
    return AMX_OK;
}

extern "C" int amx_internal_gmm_presel_info(const void* p, int* n_clusters, uint32_t* cluster_of, float* cluster_means) {
    const amx::GmmPresel* s = (const amx::GmmPresel*)p;
    if (n_clusters)
        *n_clusters = s->n_clusters;
    if (cluster_of)
        memcpy(cluster_of, s->h_cluster_of.data(), s->nk * 4);
    if (cluster_means)
        memcpy(cluster_means, s->h_cm.data(), s->h_cm.size() * 4);
    return AMX_OK;
}

static int presel_score_chunk(amx::GmmPresel* s, amx_ctx* ctx, const float* feats_dev, int T, float* scores_dev, const uint32_t* d_mix_off,
                              const uint32_t* d_k_mean, const float* d_k_const, const float* d_smeans, const float* d_isr0, int n_mix);

// frames in chunks of 16 384 like every other scorer: the distance / mask scratch stays bounded (16 MB for 256 clusters) and is
// allocated once instead of following the largest batch ever seen with a synchronising hipFree / hipMalloc
extern "C" int amx_internal_gmm_presel_score(void* p, amx_ctx* ctx, const float* feats_dev, int T, float* scores_dev, const uint32_t* d_mix_off,
                                             const uint32_t* d_k_mean, const float* d_k_const, const float* d_smeans, const float* d_isr0,
                                             int n_mix) {
    amx::GmmPresel* s = (amx::GmmPresel*)p;
    constexpr int   kChunk = 16384;
    for (int t0 = 0; t0 < T; t0 += kChunk) {
        const int r = presel_score_chunk(s, ctx, feats_dev + (size_t)t0 * s->dim, std::min(kChunk, T - t0), scores_dev + (size_t)t0 * n_mix,
                                         d_mix_off, d_k_mean, d_k_const, d_smeans, d_isr0, n_mix);
        if (r != AMX_OK)
            return r;
    }
    return AMX_OK;
}

static int presel_score_chunk(amx::GmmPresel* s, amx_ctx* ctx, const float* feats_dev, int T, float* scores_dev, const uint32_t* d_mix_off,
                              const uint32_t* d_k_mean, const float* d_k_const, const float* d_smeans, const float* d_isr0, int n_mix) {
    using namespace amx;
    AMX_HIP(hipSetDevice(ctx->device));
    const int Tpad = (T + 63) / 64 * 64, n_groups = Tpad / 64;
    if (Tpad > s->cap_T) {
        hipFree(s->d_dist);
        hipFree(s->d_masks);
        s->d_dist  = nullptr;
        s->d_masks = nullptr;
        s->cap_T   = 0;
        AMX_HIP(hipMalloc((void**)&s->d_dist, (size_t)s->n_clusters * Tpad * 4));
        AMX_HIP(hipMalloc((void**)&s->d_masks, (size_t)n_groups * s->n_clusters * 8));
        s->cap_T = Tpad;
    }
    {
        ScopedKernelTimer timer(ctx, "gmm_cluster_select");
        switch (s->dim) {
#define X(D)                                                                                                                             \
    case D:                                                                                                                              \
        hipLaunchKernelGGL((s->fma ? cluster_select_kernel<D, true> : cluster_select_kernel<D, false>), dim3(n_groups), dim3(64), 0, ctx->stream, feats_dev, d_isr0, T, Tpad, s->d_cm,     \
                           s->n_clusters, s->n_select, s->d_dist, s->d_masks, s->dim);                                                   \
        break;
            AMX_PRESEL_DIMS(X)
#undef X
            default:
                hipLaunchKernelGGL((s->fma ? cluster_select_kernel<0, true> : cluster_select_kernel<0, false>), dim3(n_groups), dim3(64), 0, ctx->stream, feats_dev, d_isr0, T, Tpad, s->d_cm,
                                   s->n_clusters, s->n_select, s->d_dist, s->d_masks, s->dim);
                break;
        }
        AMX_HIP(hipGetLastError());
    }
    int mt = 16;
    while (mt > 4 && (long)ceil_div(n_mix, mt) * ceil_div(T, 256) < 2048)
        mt /= 2;
    ScopedKernelTimer timer(ctx, "gmm");
    const dim3        grid(ceil_div(n_mix, mt), ceil_div(T, 256));
    switch (s->dim) {
#define X(D)                                                                                                                             \
    case D:                                                                                                                              \
        hipLaunchKernelGGL((s->fma ? presel_score_kernel<D, true> : presel_score_kernel<D, false>), grid, dim3(256), 0, ctx->stream, feats_dev, scores_dev, d_mix_off, d_k_mean, d_k_const, \
                           d_smeans, d_isr0, s->d_cluster_of, s->d_masks, s->n_clusters, s->backoff, T, n_mix, mt, s->dim);              \
        break;
        AMX_PRESEL_DIMS(X)
#undef X
        default:
            hipLaunchKernelGGL((s->fma ? presel_score_kernel<0, true> : presel_score_kernel<0, false>), grid, dim3(256), 0, ctx->stream, feats_dev, scores_dev, d_mix_off, d_k_mean, d_k_const,
                               d_smeans, d_isr0, s->d_cluster_of, s->d_masks, s->n_clusters, s->backoff, T, n_mix, mt, s->dim);
            break;
    }
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}
