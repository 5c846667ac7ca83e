/*
 * oracle/orc_score.c -- CPU restatement of the emission scorers (TEST INFRASTRUCTURE, see orc.h).
 *
 *   Mm::GaussDiagonalMaximumFeatureScorer / GaussDiagonalSumFeatureScorer
 *       (Mm/GaussDiagonalMaximumFeatureScorer.cc:64-86,116-298)
 *   Mm::BatchFloatFeatureScorer arithmetic (Mm/BatchFeatureScorer.cc:164-254)
 *   Nn::BatchFeatureScorer forward (Nn/BatchFeatureScorer.cc:148-171, Nn/LinearLayer.cc:298-324,
 *       Nn/ActivationLayer.cc:272-282, Nn/LinearAndActivationLayer.hh:137-160)
 *
 * Compile with -ffp-contract=off in both flavours (orc.h): the reference built for plain x86-64 (-msse3) has no fused
 * multiply-add; its default build (-march=native on an FMA host) fuses exactly the sites written ORC_FMAF / ORC_FMA here.
 */
#include "orc.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct orc_gmm {
    int       dim, n_mix, n_dens, n_mean, n_cov;
    uint32_t *mix_off, *dens_index, *dens_mean, *dens_cov;
    float*    m2lw;    /* [sum K] */
    float*    means;   /* [n_mean x dim] */
    float*    isr;     /* [n_cov x dim] */
    float*    lognorm; /* [n_cov] */
};

orc_gmm* orc_gmm_create(const orc_gmm_model* m) {
    orc_gmm* h = (orc_gmm*)calloc(1, sizeof *h);
    h->dim     = m->dim;
    h->n_mix   = m->n_mix;
    h->n_dens  = m->n_dens;
    h->n_mean  = m->n_mean;
    h->n_cov   = m->n_cov;
    size_t nk  = m->mix_offsets[m->n_mix];
    h->mix_off = (uint32_t*)malloc((size_t)(m->n_mix + 1) * 4);
    memcpy(h->mix_off, m->mix_offsets, (size_t)(m->n_mix + 1) * 4);
    h->dens_index = (uint32_t*)malloc(nk * 4);
    memcpy(h->dens_index, m->dens_index, nk * 4);
    h->dens_mean = (uint32_t*)malloc((size_t)m->n_dens * 4);
    memcpy(h->dens_mean, m->dens_mean, (size_t)m->n_dens * 4);
    h->dens_cov = (uint32_t*)malloc((size_t)m->n_dens * 4);
    memcpy(h->dens_cov, m->dens_cov, (size_t)m->n_dens * 4);
    h->means = (float*)malloc((size_t)m->n_mean * m->dim * 4);
    memcpy(h->means, m->means, (size_t)m->n_mean * m->dim * 4);

    /* Mm/MixtureFeatureScorerElement.cc:21-33: -2*logw in f64 -> f32, then f32 * scale */
    h->m2lw = (float*)malloc(nk * 4);
    for (size_t k = 0; k < nk; ++k) {
        float v    = (float)(-2 * m->log_weight[k]);
        h->m2lw[k] = v * (float)m->mixture_weight_scale; /* mixtureWeightScale_ is a Score (f32) */
    }
    /* GaussDiagonalMaximumFeatureScorer ctor (:46-62): gaussianScale_(std::sqrt(paramGaussianScale(c))): the parameter is f64,
     * std::sqrt(double), narrowed to the f32 member */
    float gs = (float)sqrt(m->gaussian_scale);
    /* Mm/CovarianceFeatureScorerElement.cc:21-51, Mm/Utilities.hh:53-91 */
    h->isr     = (float*)malloc((size_t)m->n_cov * m->dim * 4);
    h->lognorm = (float*)malloc((size_t)m->n_cov * 4);
    for (int c = 0; c < m->n_cov; ++c) {
        const float* var = m->variances + (size_t)c * m->dim;
        double       ln  = 0;
        for (int i = 0; i < m->dim; ++i) {
            /* (T)1 / (T)sqrt(x): sqrt resolves to the double overload, rounded back to f32 */
            float r                        = (float)1 / (float)sqrt((double)var[i]);
            h->isr[(size_t)c * m->dim + i] = r * gs;
            /* logNorm: log(double) of the f32 value, f64 accumulation */
            ln += log((double)fabsf(var[i]));
        }
        double g      = ORC_FMA((double)m->dim, log((double)2 * M_PI), ln); /* gaussLogNormFactor: N * log(2 pi) + logNorm, one vfmadd in the native build */
        float  f      = (float)g;
        h->lognorm[c] = f * (gs * gs); /* logNormalizationFactor_ *= factor * factor (f32) */
    }
    return h;
}

void orc_gmm_destroy(orc_gmm* h) {
    if (!h)
        return;
    free(h->mix_off);
    free(h->dens_index);
    free(h->dens_mean);
    free(h->dens_cov);
    free(h->m2lw);
    free(h->means);
    free(h->isr);
    free(h->lognorm);
    free(h);
}

int orc_contract(void) {
#ifdef ORC_CONTRACT_FMA
    return 1;
#else
    return 0;
#endif
}

const float* orc_gmm_minus2_log_weights(const orc_gmm* h) { return h->m2lw; }
const float* orc_gmm_inv_sqrt_var(const orc_gmm* h) { return h->isr; }
const float* orc_gmm_log_norm(const orc_gmm* h) { return h->lognorm; }

/* GaussDiagonalMaximumFeatureScorer::distance, __SSE3__ branch (:144-180): four strided f32
 * partial sums, hadd => (l0+l1),(l2+l3); result = 0 + ((l0+l1)+(l2+l3)); scalar tail.
 * `sum += df * df` and the tail's `result += df * df` are the contracted sites (native build: vfmadd231ps, vfmadd231ss);
 * pinned in both flavours by the function-text pin ref_gdm_distance (oracle/ref/extract_fn.py, tests/test_contract.py). */
static float orc_distance(const float* x, const float* mu, const float* isr, int dim) {
    float l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    int   eff = dim & ~3;
    int   i   = 0;
    for (; i < eff; i += 4) {
        float d0 = (mu[i] - x[i]) * isr[i];
        float d1 = (mu[i + 1] - x[i + 1]) * isr[i + 1];
        float d2 = (mu[i + 2] - x[i + 2]) * isr[i + 2];
        float d3 = (mu[i + 3] - x[i + 3]) * isr[i + 3];
        l0       = ORC_FMAF(d0, d0, l0);
        l1       = ORC_FMAF(d1, d1, l1);
        l2       = ORC_FMAF(d2, d2, l2);
        l3       = ORC_FMAF(d3, d3, l3);
    }
    float h01    = l0 + l1;
    float h23    = l2 + l3;
    float result = 0;
    result       = result + (h01 + h23);
    for (; i < dim; ++i) {
        float df = (mu[i] - x[i]) * isr[i];
        result   = ORC_FMAF(df, df, result);
    }
    return result;
}

float orc_gmm_distance(const float* x, const float* mu, const float* isr, int dim) { return orc_distance(x, mu, isr, dim); }

void orc_gmm_score(const orc_gmm* h, int mode, const float* feats, int T, float* scores, uint32_t* best) {
    int    maxk = 0;
    for (int m = 0; m < h->n_mix; ++m) {
        int k = (int)(h->mix_off[m + 1] - h->mix_off[m]);
        if (k > maxk)
            maxk = k;
    }
    float* sk = (float*)malloc((size_t)(maxk > 0 ? maxk : 1) * 4);
    for (int t = 0; t < T; ++t) {
        const float* x = feats + (size_t)t * h->dim;
        for (int m = 0; m < h->n_mix; ++m) {
            uint32_t k0 = h->mix_off[m], k1 = h->mix_off[m + 1];
            if (mode == 0) {
                /* calculateScoreAndDensity (:116-141): f64 combine, f32 running best,
                 * strict '>' so the first minimum wins */
                float    bestScore = FLT_MAX;
                uint32_t bestDns   = UINT32_MAX;
                for (uint32_t k = k0; k < k1; ++k) {
                    uint32_t d    = h->dens_index[k];
                    uint32_t c    = h->dens_cov[d];
                    float    dist = orc_distance(x, h->means + (size_t)h->dens_mean[d] * h->dim,
                                                 h->isr + (size_t)c * h->dim, h->dim);
                    double   s    = (double)h->m2lw[k] + (double)h->lognorm[c] + (double)dist;
                    if ((double)bestScore > s) {
                        bestScore = (float)s;
                        bestDns   = k - k0;
                    }
                }
                scores[(size_t)t * h->n_mix + m] = (float)(0.5 * bestScore);
                if (best)
                    best[(size_t)t * h->n_mix + m] = bestDns;
            }
            else {
                /* GaussDiagonalSumFeatureScorer (:252-298): all f32 */
                for (uint32_t k = k0; k < k1; ++k) {
                    uint32_t d     = h->dens_index[k];
                    uint32_t c     = h->dens_cov[d];
                    float    dist  = orc_distance(x, h->means + (size_t)h->dens_mean[d] * h->dim,
                                                  h->isr + (size_t)c * h->dim, h->dim);
                    float    score = h->m2lw[k] + h->lognorm[c] + dist;
                    sk[k - k0]     = (float)(0.5 * score);
                }
                float    bestScore = FLT_MAX;
                uint32_t bestDns   = UINT32_MAX;
                for (uint32_t k = 0; k < k1 - k0; ++k)
                    if (bestScore > sk[k]) {
                        bestScore = sk[k];
                        bestDns   = k;
                    }
                float sumExp = 0;
                for (uint32_t k = 0; k < k1 - k0; ++k)
                    sumExp += expf(bestScore - sk[k]);
                scores[(size_t)t * h->n_mix + m] = bestScore - logf(sumExp);
                if (best)
                    best[(size_t)t * h->n_mix + m] = bestDns;
            }
        }
    }
    free(sk);
}

/* Mm/BatchFeatureScorer.cc:207-253 (BatchFloatFeatureScorer::fillScoreCacheTpl) for ONE feature vector and one mixture: means ms
 * [nk x pdim] and the feature xs [pdim] already multiplied by 1 / sigma, pdim a multiple of 8, constants cst [nk].  Two 4-lane
 * accumulators over 8-wide blocks, lane 0 of the first starts at the constant; s1 + s2, then lanes (3 + 1) + (2 + 0); the score
 * starts at FLT_MAX and takes _mm_min_ps(score, s) = (score < s ? score : s) per density -- a NaN sum REPLACES the score, a later finite
 * one replaces the NaN --; 0.5 x where the result is below FLT_MAX.  The accumulate is `_mm_add_ps(s, _mm_mul_ps(x, x))`: vector
 * arithmetic to GCC, fused in the reference's default build (two vfmadd in the loop).  PINNED on the reference's function text in both
 * builds (oracle/ref/extract_fn.py batch_float_fill, tests/test_contract.py). */
float orc_batch_float_fill(const float* ms, const float* cst, int nk, const float* xs, int pdim) {
    float score = FLT_MAX;
    for (int k = 0; k < nk; ++k) {
        const float* mu    = ms + (size_t)k * pdim;
        float        s1[4] = {cst[k], 0, 0, 0}, s2[4] = {0, 0, 0, 0};
        for (int d = 0; d < pdim; d += 8)
            for (int j = 0; j < 4; ++j) {
                float x1 = mu[d + j] - xs[d + j];
                s1[j]    = ORC_FMAF(x1, x1, s1[j]);
                float x2 = mu[d + 4 + j] - xs[d + 4 + j];
                s2[j]    = ORC_FMAF(x2, x2, s2[j]);
            }
        float a0 = s1[0] + s2[0], a1 = s1[1] + s2[1], a2 = s1[2] + s2[2], a3 = s1[3] + s2[3];
        float r = (a3 + a1) + (a2 + a0);
        score   = score < r ? score : r; /* _mm_min_ps(score, r) */
    }
    if (score < FLT_MAX)
        score = (float)(score * 0.5);
    return score;
}

/* Mm/BatchFeatureScorer.cc:164-254 (BatchFloatFeatureScorer, "batch-diagonal-maximum-float",
 * pooled covariance only).  init(): means and features are multiplied by 1/sigma (f32), the
 * per-density constant is (f32)(logNorm - 2*logw) with the subtraction in f64 (no weight or
 * gaussian scale).  fillScoreCacheTpl(): two 4-lane f32 accumulators over 8-wide blocks,
 * lane 0 of the first one starts at the constant; a = s1+s2; result = (a3+a1)+(a2+a0);
 * min over densities; times 0.5.  log_weight must be the model's f64 log weights. */
int orc_gmm_score_batch_float(const orc_gmm* h, const double* log_weight, const float* variances,
                              const float* feats, int T, float* scores) {
    if (h->n_cov != 1)
        return -1;
    int    dim  = h->dim;
    int    pdim = ((dim + 7) / 8) * 8;
    size_t nk   = h->mix_off[h->n_mix];
    float* isr  = (float*)calloc((size_t)pdim, 4);
    double ln   = 0;
    for (int i = 0; i < dim; ++i) {
        isr[i] = (float)1 / (float)sqrt((double)variances[i]);
        ln += log((double)fabsf(variances[i]));
    }
    float  lognorm = (float)ORC_FMA((double)dim, log((double)2 * M_PI), ln);
    float* xs      = (float*)calloc((size_t)pdim, 4);
    float* ms      = (float*)calloc(nk * (size_t)pdim, 4);
    float* cst     = (float*)calloc(nk, 4);
    for (size_t k = 0; k < nk; ++k) {
        const float* mu = h->means + (size_t)h->dens_mean[h->dens_index[k]] * dim;
        for (int i = 0; i < dim; ++i)
            ms[k * pdim + i] = mu[i] * isr[i];
        cst[k] = (float)((double)lognorm - 2 * log_weight[k]);
    }
    for (int t = 0; t < T; ++t) {
        for (int i = 0; i < dim; ++i)
            xs[i] = feats[(size_t)t * dim + i] * isr[i];
        for (int m = 0; m < h->n_mix; ++m)
            scores[(size_t)t * h->n_mix + m] = orc_batch_float_fill(ms + (size_t)h->mix_off[m] * pdim, cst + h->mix_off[m],
                                                                    (int)(h->mix_off[m + 1] - h->mix_off[m]), xs, pdim);
    }
    free(isr);
    free(xs);
    free(ms);
    free(cst);
    return 0;
}

/* Mm::BatchPreselectionFloatFeatureScorer ("preselection-batch-float", Mm/BatchFeatureScorer.cc:256-318) =
 * BatchFloatFeatureScorer + Mm::FloatDensityClustering (Mm/DensityClustering.hh/.tcc):
 *   build (tcc:130-163)       k-means over the PRE-SCALED, zero-padded density means (dimension = padded dimension):
 *     initializeClusters      srand(1); cluster c starts at density rand() % nDensities (redrawn while already used)
 *     assignDensities         first cluster with the smallest unrolledVectorDistance (f32, sequential sum of squares, strict '<')
 *     updateClusterMeans      f64 component sums in density order / count -> f32; a cluster without densities keeps its mean
 *     `iterations` times (parameter default 5); nClusters is reduced to nDensities when there are fewer densities
 *   selectClusters (tcc:165-186) per frame: distance of the scaled feature to every cluster mean, ascending std::sort on the
 *     distance, the first nSelected clusters are active
 *   fillScoreCache (cc:305-318) batch-float minimum over the densities whose cluster is active; a mixture without an active
 *     density scores backoffScore (default 40000) -- the 0.5 factor is applied to real minima only.
 * cluster_of [nk] and cluster_means [n_clusters x pdim] (both nullable) return the clustering.  Equal distances in the sort are
 * broken by cluster index here (std::sort leaves their order unspecified). */
typedef struct {
    float    d;
    uint32_t c;
} orc_cl_item;

static int orc_cl_cmp(const void* a, const void* b) {
    const orc_cl_item *x = (const orc_cl_item*)a, *y = (const orc_cl_item*)b;
    if (x->d < y->d)
        return -1;
    if (x->d > y->d)
        return 1;
    return x->c < y->c ? -1 : (x->c > y->c ? 1 : 0);
}

static float orc_seq_distance(const float* a, const float* b, int dim) { /* unrolledVectorDistance<f32, f32>, dim % 8 == 0 */
    float score = 0;
    for (int i = 0; i < dim; ++i) {
        float df = a[i] - b[i];
        score    = ORC_FMAF(df, df, score); /* score += df * df: one vfmadd231ss per term in the default build (read off the disassembly of
                                             * Mm::unrolledVectorDistance<float, float> in oracle/_ref/libref_native.so; round 6) */
    }
    return score;
}

/* Mm::DensityClustering::selectClusters (Mm/DensityClustering.tcc:157-180): the n_select clusters closest to the (scaled, padded) feature.
 * The reference sorts (distance, cluster) pairs by distance with std::sort; clusters at exactly equal distances are taken here in index
 * order (the order std::sort leaves them in is the library's; equal f32 distances to two different cluster means do not occur on
 * continuous data).  sel [n_clusters]: 1 = selected.  PINNED on the reference's function text (oracle/ref/extract_fn.py
 * density_clustering), x86-64 build. */
void orc_cluster_select(const float* cm, int n_clusters, int pdim, int n_select, const float* xs, unsigned char* sel) {
    orc_cl_item* items = (orc_cl_item*)calloc((size_t)n_clusters, sizeof(orc_cl_item));
    for (int c = 0; c < n_clusters; ++c) {
        items[c].d = orc_seq_distance(xs, cm + (size_t)c * pdim, pdim);
        items[c].c = (uint32_t)c;
    }
    qsort(items, (size_t)n_clusters, sizeof(orc_cl_item), orc_cl_cmp);
    memset(sel, 0, (size_t)n_clusters);
    for (int i = 0; i < n_select; ++i)
        sel[items[i].c] = 1;
    free(items);
}

int orc_gmm_score_preselection_float(const orc_gmm* h, const double* log_weight, const float* variances, const float* feats, int T,
                                     int n_clusters, int n_select, int iterations, float backoff, float* scores,
                                     uint32_t* cluster_of_out, float* cluster_means_out, int* n_clusters_out) {
    if (h->n_cov != 1)
        return -1;
    int    dim  = h->dim;
    int    pdim = ((dim + 7) / 8) * 8;
    size_t nk   = h->mix_off[h->n_mix];
    if ((size_t)n_clusters > nk)
        n_clusters = (int)nk;
    if (n_select > n_clusters || n_select < 1)
        return -2;
    float* isr = (float*)calloc((size_t)pdim, 4);
    double ln  = 0;
    for (int i = 0; i < dim; ++i) {
        isr[i] = (float)1 / (float)sqrt((double)variances[i]);
        ln += log((double)fabsf(variances[i]));
    }
    float  lognorm = (float)ORC_FMA((double)dim, log((double)2 * M_PI), ln);
    float* xs      = (float*)calloc((size_t)pdim, 4);
    float* ms      = (float*)calloc(nk * (size_t)pdim, 4);
    float* cst     = (float*)calloc(nk, 4);
    for (size_t k = 0; k < nk; ++k) {
        const float* mu = h->means + (size_t)h->dens_mean[h->dens_index[k]] * dim;
        for (int i = 0; i < dim; ++i)
            ms[k * pdim + i] = mu[i] * isr[i];
        cst[k] = (float)((double)lognorm - 2 * log_weight[k]);
    }
    /* ---- clustering */
    float*    cm   = (float*)calloc((size_t)n_clusters * pdim, 4);
    uint32_t* cof  = (uint32_t*)calloc(nk, 4);
    char*     used = (char*)calloc(nk, 1);
    srand(1);
    for (int c = 0; c < n_clusters; ++c) {
        uint32_t pick;
        do {
            pick = (uint32_t)rand() % (uint32_t)nk;
        } while (used[pick]);
        used[pick] = 1;
        memcpy(cm + (size_t)c * pdim, ms + (size_t)pick * pdim, (size_t)pdim * 4);
    }
    double*   sums = (double*)calloc((size_t)pdim, 8);
    for (int it = 0; it < iterations; ++it) {
        for (size_t k = 0; k < nk; ++k) {
            float    bd = FLT_MAX;
            uint32_t bc = 0;
            for (int c = 0; c < n_clusters; ++c) {
                float d = orc_seq_distance(cm + (size_t)c * pdim, ms + k * pdim, pdim);
                if (d < bd) {
                    bd = d;
                    bc = (uint32_t)c;
                }
            }
            cof[k] = bc;
        }
        for (int c = 0; c < n_clusters; ++c) {
            size_t cnt = 0;
            for (int i = 0; i < pdim; ++i)
                sums[i] = 0;
            for (size_t k = 0; k < nk; ++k)
                if (cof[k] == (uint32_t)c) {
                    for (int i = 0; i < pdim; ++i)
                        sums[i] = sums[i] + (double)ms[k * pdim + i];
                    ++cnt;
                }
            if (cnt)
                for (int i = 0; i < pdim; ++i)
                    cm[(size_t)c * pdim + i] = (float)(sums[i] / (double)cnt);
        }
    }
    /* ---- scoring */
    char* active = (char*)calloc((size_t)n_clusters, 1);
    for (int t = 0; t < T; ++t) {
        for (int i = 0; i < dim; ++i)
            xs[i] = feats[(size_t)t * dim + i] * isr[i];
        orc_cluster_select(cm, n_clusters, pdim, n_select, xs, (unsigned char*)active);
        for (int m = 0; m < h->n_mix; ++m) {
            float best = FLT_MAX;
            for (uint32_t k = h->mix_off[m]; k < h->mix_off[m + 1]; ++k) {
                if (!active[cof[k]])
                    continue;
                const float* mu    = ms + (size_t)k * pdim;
                float        s1[4] = {cst[k], 0, 0, 0}, s2[4] = {0, 0, 0, 0};
                for (int d = 0; d < pdim; d += 8)
                    for (int j = 0; j < 4; ++j) {
                        float x1 = mu[d + j] - xs[d + j];
                        s1[j]    = ORC_FMAF(x1, x1, s1[j]); /* _mm_add_ps(s1, _mm_mul_ps(x1, x1)): contracted by rule (TU needs boost) */
                        float x2 = mu[d + 4 + j] - xs[d + 4 + j];
                        s2[j]    = ORC_FMAF(x2, x2, s2[j]);
                    }
                float a0 = s1[0] + s2[0], a1 = s1[1] + s2[1], a2 = s1[2] + s2[2], a3 = s1[3] + s2[3];
                float r = (a3 + a1) + (a2 + a0);
                best    = best < r ? best : r; /* _mm_min_ps(score, r): a NaN sum replaces the score (orc_batch_float_fill) */
            }
            if (best < FLT_MAX)
                best = (float)(best * 0.5);
            else if (best == FLT_MAX) /* Mm/BatchFeatureScorer.cc:310-314: `if (s == max) s = backoff` -- a NaN score stays NaN */
                best = backoff;
            scores[(size_t)t * h->n_mix + m] = best;
        }
    }
    if (cluster_of_out)
        memcpy(cluster_of_out, cof, nk * 4);
    if (cluster_means_out)
        memcpy(cluster_means_out, cm, (size_t)n_clusters * pdim * 4);
    if (n_clusters_out)
        *n_clusters_out = n_clusters;
    free(isr); free(xs); free(ms); free(cst); free(cm); free(cof); free(used); free(sums); free(active);
    return 0;
}

/* Mm::SimdGaussDiagonalMaximumFeatureScorer ("SIMD-diagonal-maximum", Mm/SimdFeatureScorer.cc:68-176) with
 * Mm::FeatureScorerIntelOptimization (Mm/IntelOptimization.cc:37-66) and quantize<f32,u8> (Mm/Utilities.hh:190-202):
 *   init (:68-82)        scaling = quantizationScalingFactor(min, max of mean * 1/sigma over all densities) (:112-137):
 *                        (f32)255 / (1.25 * 2*max(|min|,|max|)); scalingSquared = scaling^2 (f32); every covariance element
 *                        is scaled: 1/sigma *= scaling, logNorm *= scaling*scaling (Mm/CovarianceFeatureScorerElement.cc:46-52)
 *   per density (:84-109) constantWeight = (s32)((f32)((f64)(scalingSquared * -2) * logWeight) + logNorm);
 *                        preparedMean[i] = quantize(mean[i] * scaled 1/sigma[i]) = clip((int)round(v) + 128, 0, 255), zero padded
 *   per frame (:22-35)   one quantised feature vector per covariance: quantize(x[i] * scaled 1/sigma_c[i])
 *   score (:139-176)     min over the mixture's densities of constantWeight + sum (mean - feature)^2 in int, strict '<' (first
 *                        minimum); the SSE2 routine (Mm/SSE2CodeGenerator.cc: psubusb both ways, punpck, pmaddwd, paddd) is the
 *                        exact integer sum; score = (f32)(0.5 * min / scalingSquared) in f64; neither scale parameter is used.
 * (int)round(v) outside the int range is undefined in C; like the reference's x86-64 build this code relies on cvttsd2si
 * returning INT_MIN. */
static uint8_t orc_quantize_u8(float v) {
    int q = (int)round(v) + 128;
    if (q < 0)
        q = 0;
    if (q > 255)
        q = 255;
    return (uint8_t)q;
}

/* variant 1: Mm::BatchIntFeatureScorer ("batch-diagonal-maximum-int", Mm/BatchFeatureScorer.cc:338-504; "-fast" is the same
 * arithmetic unrolled): pooled covariance only; quantizationScale (:352-373) is getScaling's formula on the unscaled 1/sigma;
 * scale_ = (f32)(2.0 * scale^2) (:389); constants (s32)(logNorm * scale^2 - scale_ * logWeight) with the subtraction in f64 (:407);
 * the SSE2 distance (:427-447) is the exact integer sum; score = (f32)min / scale_ in f32 (:500); no density assignment. */
/* Mm::DensityClustering<u8, s32> (Mm/DensityClustering.tcc:61-119; PINNED on the function text, oracle/ref/extract_fn.py
 * density_clustering): srand(1) / rand() initialisation with distinct entries, `iterations` x (assign every entry to the cluster at the
 * smallest integer distance, first on ties; cluster mean = f64 sum of its entries / count, converted to u8).  means [nk x dim] per mixture
 * entry; cof [nk], cm [n_clusters x dim]. */
static int orc_int_distance(const uint8_t* a, const uint8_t* b, int dim);
void orc_cluster_u8(const uint8_t* means, int nk, int dim, int n_clusters, int iterations, uint32_t* cof, uint8_t* cm) {
    char* used = (char*)calloc((size_t)nk, 1);
    srand(1);
    for (int c = 0; c < n_clusters; ++c) {
        uint32_t pick;
        do {
            pick = (uint32_t)rand() % (uint32_t)nk;
        } while (used[pick]);
        used[pick] = 1;
        memcpy(cm + (size_t)c * dim, means + (size_t)pick * dim, (size_t)dim);
    }
    free(used);
    double* sums = (double*)calloc((size_t)dim, 8);
    for (int it = 0; it < iterations; ++it) {
        for (int k = 0; k < nk; ++k) {
            int      bd = INT32_MAX;
            uint32_t bc = 0;
            for (int c = 0; c < n_clusters; ++c) {
                int d = orc_int_distance(cm + (size_t)c * dim, means + (size_t)k * dim, dim);
                if (d < bd) {
                    bd = d;
                    bc = (uint32_t)c;
                }
            }
            cof[k] = bc;
        }
        for (int c = 0; c < n_clusters; ++c) {
            size_t cnt = 0;
            for (int i = 0; i < dim; ++i)
                sums[i] = 0;
            for (int k = 0; k < nk; ++k)
                if (cof[k] == (uint32_t)c) {
                    for (int i = 0; i < dim; ++i)
                        sums[i] = sums[i] + (double)means[(size_t)k * dim + i];
                    ++cnt;
                }
            if (cnt)
                for (int i = 0; i < dim; ++i)
                    cm[(size_t)c * dim + i] = (uint8_t)(sums[i] / (double)cnt);
        }
    }
    free(sums);
}

/* variant 1 with `ps`: Mm::BatchPreselectionIntFeatureScorer ("preselection-batch-int", Mm/BatchFeatureScorer.cc:514-578) =
 * the batch-int scorer restricted to the densities of the nSelected clusters closest to the quantised feature, with
 * Mm::DensityClustering<u8, s32> (Mm/DensityClustering.tcc, see the float variant above) over the quantised per-entry means:
 * s32 sums of squared differences (Mm/Utilities.hh unrolledVectorDistance<u8, s32>), strict '<', f64 component sums / count
 * converted to u8 (truncation); a mixture without an active density keeps best = INT_MAX and scores (f32)INT_MAX / scale_
 * (fillScoreCacheTpl, :458-499: no back-off score in the int class). */
typedef struct {
    int       n_clusters, n_select, iterations;
    uint32_t* cluster_of_out;     /* [nk], nullable */
    uint8_t*  cluster_means_out;  /* [n_clusters x dim], nullable */
    int*      n_clusters_out;
} orc_presel_int;

typedef struct {
    int      d;
    uint32_t c;
} orc_cli_item;

static int orc_cli_cmp(const void* a, const void* b) {
    const orc_cli_item *x = (const orc_cli_item*)a, *y = (const orc_cli_item*)b;
    if (x->d != y->d)
        return x->d < y->d ? -1 : 1;
    return x->c < y->c ? -1 : (x->c > y->c ? 1 : 0);
}

static int orc_int_distance(const uint8_t* a, const uint8_t* b, int dim) {
    int s = 0;
    for (int i = 0; i < dim; ++i) {
        int df = (int)a[i] - (int)b[i];
        s += df * df;
    }
    return s;
}

static int orc_gmm_score_quantized(const orc_gmm* h, int variant, const double* log_weight, const float* variances, const float* feats,
                                   int T, float* scores, uint32_t* best, float* scaling_out, orc_presel_int* ps) {
    if (variant == 1 && h->n_cov != 1)
        return -1;
    const int dim = h->dim;
    size_t    nk  = h->mix_off[h->n_mix];
    float*    isr = (float*)malloc((size_t)h->n_cov * dim * 4);
    float*    lognorm = (float*)malloc((size_t)h->n_cov * 4);
    for (int c = 0; c < h->n_cov; ++c) {
        double ln = 0;
        for (int i = 0; i < dim; ++i) {
            float v                 = variances[(size_t)c * dim + i];
            isr[(size_t)c * dim + i] = (float)1 / (float)sqrt((double)v);
            ln += log((double)fabsf(v));
        }
        lognorm[c] = (float)ORC_FMA((double)dim, log((double)2 * M_PI), ln); /* gaussLogNormFactor, as above (one vfmadd in the native build) */
    }
    float minMean = FLT_MAX, maxMean = -FLT_MAX;
    for (int d = 0; d < h->n_dens; ++d) {
        const float* mu = h->means + (size_t)h->dens_mean[d] * dim;
        const float* is = isr + (size_t)h->dens_cov[d] * dim;
        for (int i = 0; i < dim; ++i) {
            float dm = mu[i] * is[i];
            minMean  = dm < minMean ? dm : minMean;
            maxMean  = maxMean < dm ? dm : maxMean;
        }
    }
    float amin = fabsf(minMean), amax = fabsf(maxMean);
    float intervalSize = 2 * (amin < amax ? amax : amin);
    float scaling      = (float)((float)255 / (1.25 * intervalSize));
    float scaling2     = scaling * scaling;
    if (scaling_out)
        *scaling_out = scaling;
    for (int c = 0; c < h->n_cov; ++c) {
        for (int i = 0; i < dim; ++i)
            isr[(size_t)c * dim + i] = isr[(size_t)c * dim + i] * scaling;
        lognorm[c] = lognorm[c] * (scaling * scaling);
    }
    uint8_t* qmean = (uint8_t*)malloc((size_t)h->n_dens * dim);
    for (int d = 0; d < h->n_dens; ++d)
        for (int i = 0; i < dim; ++i)
            qmean[(size_t)d * dim + i] = orc_quantize_u8(h->means[(size_t)h->dens_mean[d] * dim + i] * isr[(size_t)h->dens_cov[d] * dim + i]);
    int32_t* cst = (int32_t*)malloc((nk ? nk : 1) * 4);
    for (size_t k = 0; k < nk; ++k) {
        double scaledM2lw = (double)(scaling2 * -2) * log_weight[k];
        float  asScore    = (float)scaledM2lw;
        cst[k]            = (int32_t)(asScore + lognorm[h->dens_cov[h->dens_index[k]]]);
    }
    const float int_scale = (float)(2.0 * scaling2);
    if (variant == 1)
        for (size_t k = 0; k < nk; ++k)
            cst[k] = (int32_t)((double)lognorm[0] - (double)int_scale * log_weight[k]);
    /* ---- density clustering over the per-entry quantised means (preselection only) */
    uint8_t*      cm     = NULL;
    uint32_t*     cof    = NULL;
    char*         active = NULL;
    orc_cli_item* items  = NULL;
    int           n_clusters = 0;
    if (ps) {
        n_clusters = ps->n_clusters;
        if ((size_t)n_clusters > nk)
            n_clusters = (int)nk;
        if (variant != 1 || ps->n_select > n_clusters || ps->n_select < 1 || n_clusters < 1) {
            free(isr); free(lognorm); free(qmean); free(cst);
            return -2;
        }
        cm         = (uint8_t*)calloc((size_t)n_clusters * dim, 1);
        cof        = (uint32_t*)calloc(nk, 4);
        active     = (char*)calloc((size_t)n_clusters, 1);
        items      = (orc_cli_item*)calloc((size_t)n_clusters, sizeof(orc_cli_item));
        /* the clustering sees the means per mixture ENTRY, as BatchIntFeatureScorer::init lays them out */
        uint8_t* em = (uint8_t*)malloc(nk * (size_t)dim);
        for (size_t k = 0; k < nk; ++k)
            memcpy(em + k * dim, qmean + (size_t)h->dens_index[k] * dim, (size_t)dim);
        orc_cluster_u8(em, (int)nk, dim, n_clusters, ps->iterations, cof, cm);
        free(em);
        if (ps->cluster_of_out)
            memcpy(ps->cluster_of_out, cof, nk * 4);
        if (ps->cluster_means_out)
            memcpy(ps->cluster_means_out, cm, (size_t)n_clusters * dim);
        if (ps->n_clusters_out)
            *ps->n_clusters_out = n_clusters;
    }
    uint8_t* qx = (uint8_t*)malloc((size_t)h->n_cov * dim);
    for (int t = 0; t < T; ++t) {
        const float* x = feats + (size_t)t * dim;
        for (int c = 0; c < h->n_cov; ++c)
            for (int i = 0; i < dim; ++i)
                qx[(size_t)c * dim + i] = orc_quantize_u8(x[i] * isr[(size_t)c * dim + i]);
        if (ps) { /* selectClusters: ascending sort on the distance, the first nSelected are active (ties: by cluster index) */
            for (int c = 0; c < n_clusters; ++c) {
                items[c].d = orc_int_distance(qx, cm + (size_t)c * dim, dim);
                items[c].c = (uint32_t)c;
            }
            qsort(items, (size_t)n_clusters, sizeof(orc_cli_item), orc_cli_cmp);
            memset(active, 0, (size_t)n_clusters);
            for (int i = 0; i < ps->n_select; ++i)
                active[items[i].c] = 1;
        }
        for (int m = 0; m < h->n_mix; ++m) {
            int      minScore = INT32_MAX;
            uint32_t bestDns  = UINT32_MAX;
            for (uint32_t k = h->mix_off[m]; k < h->mix_off[m + 1]; ++k) {
                if (ps && !active[cof[k]])
                    continue;
                uint32_t       d  = h->dens_index[k];
                const uint8_t* a  = qmean + (size_t)d * dim;
                const uint8_t* b  = qx + (size_t)h->dens_cov[d] * dim;
                int            ds = 0;
                for (int i = 0; i < dim; ++i) {
                    int df = (int)a[i] - (int)b[i];
                    ds += df * df;
                }
                int score = cst[k] + ds;
                if (score < minScore) {
                    minScore = score;
                    bestDns  = k - h->mix_off[m];
                }
            }
            scores[(size_t)t * h->n_mix + m] = variant == 1 ? (float)minScore / int_scale : (float)(0.5 * minScore / scaling2);
            if (best)
                best[(size_t)t * h->n_mix + m] = bestDns;
        }
    }
    free(isr);
    free(lognorm);
    free(qmean);
    free(cst);
    free(qx);
    free(cm);
    free(cof);
    free(active);
    free(items);
    return 0;
}

int orc_gmm_score_simd(const orc_gmm* h, const double* log_weight, const float* variances, const float* feats, int T, float* scores,
                       uint32_t* best, float* scaling_out) {
    return orc_gmm_score_quantized(h, 0, log_weight, variances, feats, T, scores, best, scaling_out, NULL);
}

int orc_gmm_score_batch_int(const orc_gmm* h, const double* log_weight, const float* variances, const float* feats, int T,
                            float* scores) {
    return orc_gmm_score_quantized(h, 1, log_weight, variances, feats, T, scores, NULL, NULL, NULL);
}

int orc_gmm_score_preselection_int(const orc_gmm* h, const double* log_weight, const float* variances, const float* feats, int T,
                                   int n_clusters, int n_select, int iterations, float* scores, uint32_t* cluster_of_out,
                                   uint8_t* cluster_means_out, int* n_clusters_out) {
    orc_presel_int ps = {n_clusters, n_select, iterations, cluster_of_out, cluster_means_out, n_clusters_out};
    return orc_gmm_score_quantized(h, 1, log_weight, variances, feats, T, scores, NULL, NULL, &ps);
}

/* the quantiser alone, for pinning against the reference's functor */
unsigned orc_quantize(float v) { return orc_quantize_u8(v); }

/* ------------------------------------------------------------------ Viterbi accumulation (training statistics)
 * Mm::AbstractMixtureSetEstimator::accumulate(mixture, x) with viterbi_ = true
 * (Mm/AbstractMixtureSetEstimator.cc:117-125): the density with the best score of the aligned mixture gets
 * weight += 1 (Mm::MixtureEstimator), its mean accumulator sum += x, weight += 1 (std::plus<f64>), its covariance
 * accumulator sum += x*x, weight += 1 (plusSquare<f64>) (Mm/GaussDensityEstimator.hh:152-208, Mm/VectorAccumulator.hh:60-63).
 * Layout of acc (f64): [nk mixture-density weights][n_mean weights][n_mean x dim sums][n_cov weights][n_cov x dim sums]. */
long orc_gmm_accumulator_size(const orc_gmm* h) {
    return (long)h->mix_off[h->n_mix] + (long)h->n_mean * (1 + h->dim) + (long)h->n_cov * (1 + h->dim);
}

void orc_gmm_accumulate(const orc_gmm* h, const float* feats, int T, const uint32_t* mixture, const uint32_t* density_in_mixture,
                        double* acc) {
    const size_t nk = h->mix_off[h->n_mix];
    double*      mw = acc + nk;
    double*      ms = mw + h->n_mean;
    double*      cw = ms + (size_t)h->n_mean * h->dim;
    double*      cs = cw + h->n_cov;
    for (int t = 0; t < T; ++t) {
        const uint32_t k  = h->mix_off[mixture[t]] + density_in_mixture[t];
        const uint32_t d  = h->dens_index[k];
        const uint32_t mi = h->dens_mean[d], ci = h->dens_cov[d];
        const float*   x  = feats + (size_t)t * h->dim;
        acc[k] += 1;
        mw[mi] += 1;
        cw[ci] += 1;
        for (int i = 0; i < h->dim; ++i) {
            double y = x[i];
            ms[(size_t)mi * h->dim + i] = ms[(size_t)mi * h->dim + i] + y;
            cs[(size_t)ci * h->dim + i] = cs[(size_t)ci * h->dim + i] + y * y;
        }
    }
}

/* Weighted Viterbi and Baum-Welch accumulation: Mm::AbstractMixtureSetEstimator::accumulate(mixture, x, weight)
 * (Mm/AbstractMixtureSetEstimator.cc:127-147).
 *   mode 0 (viterbi): the best density of the aligned mixture gets the frame's weight: MixtureEstimator weights_[k] += w
 *     (Mm/MixtureEstimator.cc:99-104), mean sum = sum + w*x (plusWeighted<f64>, Mm/Utilities.hh:108-122), covariance
 *     sum = sum + w*x*x (plusSquareWeighted<f64>, :135-153: left to right, (w*x)*x), both weights += w
 *     (Mm/VectorAccumulator.hh:64-67).
 *   mode 1 (baum-welch): density posteriors of the log-add scorer, p_k = exp(score(e) - s_k) in f32
 *     (Mm/GaussDiagonalMaximumFeatureScorer.cc:291-298 with logDenominator = score(e), Mm/AssigningFeatureScorer.hh:139-141);
 *     final weight = w * p_k in f64; densities with final weight > Core::Type<f32>::epsilon (weightThreshold_, ctor :63)
 *     accumulate x with that weight.
 * weight == NULL means 1.0 for every frame. */
void orc_gmm_accumulate_weighted(const orc_gmm* h, int mode, const float* feats, int T, const uint32_t* mixture, const double* weight,
                                 const uint32_t* density_in_mixture, double* acc) {
    const size_t nk = h->mix_off[h->n_mix];
    double*      mw = acc + nk;
    double*      ms = mw + h->n_mean;
    double*      cw = ms + (size_t)h->n_mean * h->dim;
    double*      cs = cw + h->n_cov;
    int          maxk = 1;
    for (int m = 0; m < h->n_mix; ++m)
        if ((int)(h->mix_off[m + 1] - h->mix_off[m]) > maxk)
            maxk = (int)(h->mix_off[m + 1] - h->mix_off[m]);
    float* sk = (float*)malloc((size_t)maxk * 4);
    for (int t = 0; t < T; ++t) {
        const uint32_t m  = mixture[t];
        const uint32_t k0 = h->mix_off[m], nd = h->mix_off[m + 1] - k0;
        const float*   x  = feats + (size_t)t * h->dim;
        const double   w  = weight ? weight[t] : 1.0;
        float          logDen = 0;
        if (mode == 1) {
            for (uint32_t j = 0; j < nd; ++j) {
                uint32_t d     = h->dens_index[k0 + j];
                uint32_t c     = h->dens_cov[d];
                float    dist  = orc_distance(x, h->means + (size_t)h->dens_mean[d] * h->dim, h->isr + (size_t)c * h->dim, h->dim);
                float    score = h->m2lw[k0 + j] + h->lognorm[c] + dist;
                sk[j]          = (float)(0.5 * score);
            }
            float bestScore = FLT_MAX;
            for (uint32_t j = 0; j < nd; ++j)
                if (bestScore > sk[j])
                    bestScore = sk[j];
            float sumExp = 0;
            for (uint32_t j = 0; j < nd; ++j)
                sumExp += expf(bestScore - sk[j]);
            logDen = bestScore - logf(sumExp);
        }
        for (uint32_t j = 0; j < nd; ++j) {
            double fw;
            if (mode == 1) {
                double p = (double)expf(logDen - sk[j]);
                fw       = w * p;
                if (!(fw > (double)FLT_EPSILON))
                    continue;
            }
            else {
                if (j != density_in_mixture[t])
                    continue;
                fw = w;
            }
            const uint32_t k  = k0 + j;
            const uint32_t d  = h->dens_index[k];
            const uint32_t mi = h->dens_mean[d], ci = h->dens_cov[d];
            acc[k] += fw;
            mw[mi] += fw;
            cw[ci] += fw;
            for (int i = 0; i < h->dim; ++i) {
                double y = x[i];
                ms[(size_t)mi * h->dim + i] = ms[(size_t)mi * h->dim + i] + fw * y;
                cs[(size_t)ci * h->dim + i] = cs[(size_t)ci * h->dim + i] + fw * y * y;
            }
        }
    }
    free(sk);
}

/* ------------------------------------------------------------------ FFNN forward */

static float orc_act(float v, int act) {
    switch (act) {
        case ORC_ACT_RELU: return v < 0 ? 0 : v;                     /* ensureMinimalValue(0) */
        case ORC_ACT_SIGMOID: { /* Math/FastMatrix.hh:802-808: scale(-1), exp(), 1.0 / (1.0 + e) in f64.  exp() is mt_vr_exp
                                 * (Math/FastVectorOperations.hh:57-63): the unqualified exp(x[i]) on a float resolves to ::exp(double)
                                 * there, narrowed on assignment -- PINNED on that template compiled unmodified (ref_mt_vr_exp) */
            float e = (float)exp((double)-v);
            return (float)(1.0 / (1.0 + e));
        }
        case ORC_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

float orc_activation(float v, int act) { return orc_act(v, act); }

void orc_ffnn_score(const orc_ffnn_model* m, const float* feats, int T, float* scores, int acc64) {
    int maxd = m->in_dim[0];
    for (int l = 0; l < m->n_layers; ++l)
        if (m->out_dim[l] > maxd)
            maxd = m->out_dim[l];
    float* a = (float*)malloc((size_t)maxd * 4);
    float* b = (float*)malloc((size_t)maxd * 4);
    for (int t = 0; t < T; ++t) {
        memcpy(a, feats + (size_t)t * m->in_dim[0], (size_t)m->in_dim[0] * 4);
        for (int l = 0; l < m->n_layers; ++l) {
            int          in = m->in_dim[l], out = m->out_dim[l];
            const float* W    = m->W[l];
            int          last = (l == m->n_layers - 1);
            for (int o = 0; o < out; ++o) {
                const float* w = W + (size_t)o * in;
                float        z;
                if (acc64 == 2) { /* k-ordered fused multiply-add chain (what an f32 MFMA computes) */
                    float s = 0;
                    for (int i = 0; i < in; ++i)
                        s = fmaf(w[i], a[i], s);
                    z = s;
                }
                else if (acc64) {
                    double s = 0;
                    for (int i = 0; i < in; ++i)
                        s += (double)w[i] * (double)a[i];
                    z = (float)s;
                }
                else {
                    float s = 0;
                    for (int i = 0; i < in; ++i)
                        s = s + w[i] * a[i];
                    z = s;
                }
                float bias = m->bias[l] ? m->bias[l][o] : 0.f;
                if (last && m->log_prior && m->prior_scale != 0.f)
                    bias = bias - m->prior_scale * m->log_prior[o]; /* removeLogPriorFromBias */
                z = z + bias;                                        /* addToAllColumns */
                z = orc_act(z, m->activation[l]);
                b[o] = z;
            }
            float* tmp = a;
            a          = b;
            b          = tmp;
        }
        int outl = m->out_dim[m->n_layers - 1];
        for (int o = 0; o < outl; ++o)
            scores[(size_t)t * outl + o] = -a[o]; /* Nn/BatchFeatureScorer.cc:165 */
    }
    free(a);
    free(b);
}

/* Math::FastMatrix<f32>::softmax (Math/FastMatrix.hh:818-834) on every row of x [T x n] (a row = one frame = a column of the reference's
 * activation matrix): maximum (FastVector::getMaxOfColumns: std::max_element), x + (-1 * max) (addToAllRows), exp() = mt_vr_exp
 * (Math/FastVectorOperations.hh:57-63: ::exp(double) narrowed; pinned by ref_mt_vr_exp), the sequential f32 sum of
 * FastVector::addSummedRows (Math/FastVector.hh:481-489), scal by (f32)1.0 / sum (divideColumnsByScalars).  PARITY UNPINNED as a
 * whole (FastMatrix.hh needs the BLAS headers); the exponential is pinned. */
void orc_softmax_rows(float* x, int T, int n) {
    for (int t = 0; t < T; ++t) {
        float* r  = x + (size_t)t * n;
        float  mx = r[0];
        for (int i = 1; i < n; ++i)
            if (mx < r[i])
                mx = r[i];
        const float value = -1.f * mx;
        for (int i = 0; i < n; ++i) {
            float d = r[i] + value;
            r[i]    = (float)exp((double)d);
        }
        float sum = 0.f;
        for (int i = 0; i < n; ++i)
            sum += 1.f * r[i];
        const float inv = (float)1.0 / sum;
        for (int i = 0; i < n; ++i)
            r[i] = r[i] * inv;
    }
}

/* Nn::NeuralNetworkForwardNode (Nn/NeuralNetworkForwardNode.cc:140-180): the top layer's output per frame -- the activation
 * W x + b - alpha logPrior (top = 0; a linear+softmax layer with evaluate-softmax = false) or its softmax (top = 1, the default) */
void orc_ffnn_forward(const orc_ffnn_model* m, const float* feats, int T, float* out, int top, int acc64) {
    orc_ffnn_score(m, feats, T, out, acc64);
    const int n = m->out_dim[m->n_layers - 1];
    for (size_t i = 0; i < (size_t)T * n; ++i)
        out[i] = -out[i];
    if (top == 1)
        orc_softmax_rows(out, T, n);
}
