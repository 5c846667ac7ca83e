// pms_io.cpp -- reader / writer of RASR's text mixture-set files (".pms").
//
// Format written by Mm::MixtureSet::write(std::ostream&) (Mm/MixtureSet.cc:141-168):
//   #Version: 2.0
//   #CovarianceType: DiagonalCovariance
//   <dim> <nMixtures> <nDensities> <nMeans> <nCovariances>
//   per mixture    : <K> {<densityIndex> <logWeight>}*K            (Mm/Mixture.cc:80-87)
//   per density    : <meanIndex> <covarianceIndex>                 (Mm/MixtureSetTopology.cc:19-22)
//   per mean       : <dim> {<value>}*dim                           (Mm/GaussDensity.cc:25-31)
//   per covariance : " " <dim> {<variance> <weight>}*dim           (Mm/GaussDensity.cc:45-53)
// Reading (Mm/MixtureSet.cc:169-216): versions < 2.0 store linear weights, which become
// log(w) or Core::Type<f64>::min for w <= 0 (Mm/Mixture.cc:63-66); the diagonal is
// variance*weight with all feature weights reset to 1 (Mm/GaussDensity.cc:54-70).
//
// Also here, because it produces the same in-memory mixture set: amx_gmm_estimate, the re-estimation (M-step) and the
// mixture splitter of the acoustic model trainer.  Model-sized host work that runs once per epoch on the all-reduced
// statistics, like the reference's "estimate" / "split" actions.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.hpp"

struct amx_mixture_set {
    int                   dim = 0;
    std::vector<uint32_t> mix_off, dens_index, dens_mean, dens_cov;
    std::vector<double>   log_weight;
    std::vector<float>    means, variances;
};

extern "C" {

int amx_pms_read(const char* path, amx_mixture_set** out) {
    AMX_REQUIRE(path && out, AMX_ERR_INVALID, "amx_pms_read: NULL argument");
    *out = nullptr;
    std::ifstream in(path);
    AMX_REQUIRE(in.good(), AMX_ERR_INVALID, "amx_pms_read: cannot open '%s'", path);
    std::string line;
    std::getline(in, line);
    AMX_REQUIRE(line.size() > 10 && line.compare(0, 9, "#Version:") == 0, AMX_ERR_INVALID, "amx_pms_read: '%s' has no #Version header", path);
    const float version = (float)atof(line.substr(10).c_str());
    AMX_REQUIRE(!(version > 2.0), AMX_ERR_UNSUPPORTED, "amx_pms_read: version \"%g\" not supported", version);
    std::getline(in, line);
    AMX_REQUIRE(line.size() >= 17 && line.substr(17).compare("DiagonalCovariance") == 0, AMX_ERR_UNSUPPORTED,
                "amx_pms_read: No correct Covariance Type set: %s", line.size() >= 17 ? line.substr(17).c_str() : line.c_str());
    unsigned dim = 0, nMix = 0, nDns = 0, nMean = 0, nCov = 0;
    in >> dim >> nMix >> nDns >> nMean >> nCov;
    AMX_REQUIRE(in.good() && dim > 0, AMX_ERR_INVALID, "amx_pms_read: bad size line in '%s'", path);
    amx_mixture_set* ms = new amx_mixture_set;
    ms->dim             = (int)dim;
    ms->mix_off.push_back(0);
    // every loop stops at the first failed extraction: a corrupt count in the header must not turn into 2^32 push_backs
    for (unsigned m = 0; m < nMix && in.good(); ++m) {
        unsigned k = 0;
        in >> k;
        for (unsigned j = 0; j < k && in.good(); ++j) {
            unsigned d = 0;
            double   w = 0;
            in >> d >> w;
            if (version < 2.0)
                w = w > 0 ? std::log(w) : -1.7976931348623157e+308;
            ms->dens_index.push_back(d);
            ms->log_weight.push_back(w);
        }
        ms->mix_off.push_back((uint32_t)ms->dens_index.size());
    }
    for (unsigned d = 0; d < nDns && in.good(); ++d) {
        unsigned mi = 0, ci = 0;
        in >> mi >> ci;
        ms->dens_mean.push_back(mi);
        ms->dens_cov.push_back(ci);
    }
    for (unsigned i = 0; i < nMean && in.good(); ++i) {
        unsigned n = 0;
        in >> n;
        if (in.fail())
            break;
        if (n != dim) {
            amx::set_error("amx_pms_read: mean %u has dimension %u, expected %u", i, n, dim);
            delete ms;
            return AMX_ERR_INVALID;
        }
        for (unsigned j = 0; j < n && !in.fail(); ++j) {
            float v = 0;
            in >> v;
            ms->means.push_back(v);
        }
    }
    for (unsigned i = 0; i < nCov && !in.fail(); ++i) {
        unsigned n = 0;
        in >> n;
        if (in.fail())
            break;
        if (n != dim) {
            amx::set_error("amx_pms_read: covariance %u has dimension %u, expected %u", i, n, dim);
            delete ms;
            return AMX_ERR_INVALID;
        }
        for (unsigned j = 0; j < n && !in.fail(); ++j) {
            float  v = 0;
            double w = 0;
            in >> v >> w;
            ms->variances.push_back((float)(v * w));
        }
    }
    if (in.fail() || ms->mix_off.size() != (size_t)nMix + 1 || ms->dens_mean.size() != nDns || ms->means.size() != (size_t)nMean * dim ||
        ms->variances.size() != (size_t)nCov * dim) {
        amx::set_error("amx_pms_read: '%s' is truncated or malformed", path);
        delete ms;
        return AMX_ERR_INVALID;
    }
    *out = ms;
    return AMX_OK;
}

int amx_pms_write(const amx_gmm_model* m, const char* path) {
    AMX_REQUIRE(m && path, AMX_ERR_INVALID, "amx_pms_write: NULL argument");
    std::ofstream o(path);
    AMX_REQUIRE(o.good(), AMX_ERR_INVALID, "amx_pms_write: cannot open '%s'", path);
    // the reference streams with the default 6 significant digits; we keep full round-trip
    // precision instead so that a written model scores identically after re-reading
    o << std::setprecision(17);
    o << "#Version: 2.0" << std::endl;
    o << "#CovarianceType: DiagonalCovariance" << std::endl;
    o << m->dim << " " << m->n_mix << " " << m->n_dens << " " << m->n_mean << " " << m->n_cov << std::endl;
    for (int i = 0; i < m->n_mix; ++i) {
        o << (m->mix_offsets[i + 1] - m->mix_offsets[i]);
        for (uint32_t k = m->mix_offsets[i]; k < m->mix_offsets[i + 1]; ++k)
            o << " " << m->dens_index[k] << " " << m->log_weight[k];
        o << std::endl;
    }
    for (int d = 0; d < m->n_dens; ++d)
        o << m->dens_mean[d] << " " << m->dens_cov[d] << std::endl;
    o << std::setprecision(9);
    for (int i = 0; i < m->n_mean; ++i) {
        o << m->dim;
        for (int j = 0; j < m->dim; ++j)
            o << " " << m->means[(size_t)i * m->dim + j];
        o << std::endl;
    }
    for (int i = 0; i < m->n_cov; ++i) {
        o << " " << m->dim;
        for (int j = 0; j < m->dim; ++j)
            o << " " << m->variances[(size_t)i * m->dim + j] << " " << 1;
        o << std::endl;
    }
    AMX_REQUIRE(o.good(), AMX_ERR_INVALID, "amx_pms_write: write to '%s' failed", path);
    return AMX_OK;
}

int amx_mixture_set_view(const amx_mixture_set* ms, amx_gmm_model* v) {
    AMX_REQUIRE(ms && v, AMX_ERR_INVALID, "amx_mixture_set_view: NULL argument");
    v->dim                  = ms->dim;
    v->n_mix                = (int)ms->mix_off.size() - 1;
    v->n_dens               = (int)ms->dens_mean.size();
    v->n_mean               = (int)(ms->means.size() / (size_t)ms->dim);
    v->n_cov                = (int)(ms->variances.size() / (size_t)ms->dim);
    v->mix_offsets          = ms->mix_off.data();
    v->dens_index           = ms->dens_index.data();
    v->log_weight           = ms->log_weight.data();
    v->dens_mean            = ms->dens_mean.data();
    v->dens_cov             = ms->dens_cov.data();
    v->means                = ms->means.data();
    v->variances            = ms->variances.data();
    v->mixture_weight_scale = 1.0;
    v->gaussian_scale       = 1.0;
    v->tuning               = nullptr;
    return AMX_OK;
}

void amx_mixture_set_destroy(amx_mixture_set* ms) {
    delete ms;
}

// ------------------------------------------------------------------ re-estimation and splitting
//
// Mm::AbstractMixtureSetEstimator::estimate (Mm/AbstractMixtureSetEstimator.cc:305-338) on the flat accumulator of
// amx_gmm_accumulate*_dev, followed (cfg->split) by Mm::MixtureSetSplitter::split (Mm/MixtureSetSplitter.cc:38-123).
extern "C++" {
namespace {
// Mm/Utilities.hh:43-51 (the first maximum is left out of the sum)
double log_exp_norm(const std::vector<double>& v, size_t a, size_t b) {
    size_t mx = a;
    for (size_t i = a + 1; i < b; ++i)
        if (v[mx] < v[i])
            mx = i;
    double r = 0;
    for (size_t i = a; i < b; ++i)
        if (i != mx)
            r += std::exp(v[i] - v[mx]);
    return std::log1p(r) + v[mx];
}
// Core::ReferenceIndexMap::add: index of first appearance
uint32_t first_seen(std::unordered_map<uint32_t, uint32_t>& map, std::vector<uint32_t>& order, uint32_t key) {
    auto it = map.find(key);
    if (it != map.end())
        return it->second;
    const uint32_t idx = (uint32_t)order.size();
    map.emplace(key, idx);
    order.push_back(key);
    return idx;
}
}  // namespace
}  // extern "C++"

void amx_gmm_estimate_cfg_default(amx_gmm_estimate_cfg* c) {
    if (!c)
        return;
    c->min_observation_weight                  = 5;      // Mm/AbstractMixtureSetEstimator.cc:25-28
    c->min_relative_weight                     = 0;      // :30-33
    c->min_variance                            = 0;      // :35-38
    c->normalize_mixture_weights               = 1;      // :56-59
    c->allow_zero_weights                      = 0;      // :51-54
    c->split                                   = 0;
    c->split_min_mean_observation_weight       = 20;     // Mm/MixtureSetSplitter.cc:20-21
    c->split_min_covariance_observation_weight = FLT_MAX;  // :22-24
    c->split_perturbation_weight               = 0.1;    // :25-26
    c->split_normalize_mixture_weights         = 0;      // :27-28
}

int amx_gmm_estimate(const amx_gmm_model* t, const double* acc, const amx_gmm_estimate_cfg* cfg_in, amx_mixture_set** out) {
    AMX_REQUIRE(t && acc && out, AMX_ERR_INVALID, "amx_gmm_estimate: NULL argument");
    *out = nullptr;
    AMX_REQUIRE(t->dim > 0 && t->n_mix >= 0 && t->mix_offsets && t->dens_index && t->dens_mean && t->dens_cov, AMX_ERR_INVALID,
                "amx_gmm_estimate: incomplete topology");
    amx_gmm_estimate_cfg c;
    if (cfg_in)
        c = *cfg_in;
    else
        amx_gmm_estimate_cfg_default(&c);
    const int      dim = t->dim, n_mix = t->n_mix;
    const size_t   nk = t->mix_offsets[n_mix];
    const double*  kw = acc;
    const double*  mw = acc + nk;
    const double*  ms = mw + t->n_mean;
    const double*  cw = ms + (size_t)t->n_mean * dim;
    const double*  cs = cw + t->n_cov;
    for (size_t k = 0; k < nk; ++k)
        AMX_REQUIRE((int)t->dens_index[k] < t->n_dens, AMX_ERR_INVALID, "amx_gmm_estimate: density index out of range");
    for (int d = 0; d < t->n_dens; ++d)
        AMX_REQUIRE((int)t->dens_mean[d] < t->n_mean && (int)t->dens_cov[d] < t->n_cov, AMX_ERR_INVALID,
                    "amx_gmm_estimate: mean / covariance index out of range");

    // checkEventsWithZeroWeight (:422-431)
    if (!c.allow_zero_weights)
        for (int m = 0; m < n_mix; ++m) {
            double total = 0;
            for (uint32_t k = t->mix_offsets[m]; k < t->mix_offsets[m + 1]; ++k)
                total += kw[k];
            AMX_REQUIRE(!(total == 0 && t->mix_offsets[m + 1] > t->mix_offsets[m]), AMX_ERR_STATE, "Mixture %d has zero weight.", m);
        }

    // CovarianceToMeanSetMap over every density the mixtures reach, taken BEFORE densities are removed (:309-313).  The
    // reference walks an unordered_set keyed by pointer; mean-index order here.
    std::vector<std::vector<uint32_t>> cov_means((size_t)t->n_cov);
    {
        std::vector<char> seen((size_t)t->n_dens, 0);
        for (size_t k = 0; k < nk; ++k) {
            const uint32_t d = t->dens_index[k];
            if (!seen[d]) {
                seen[d] = 1;
                cov_means[t->dens_cov[d]].push_back(t->dens_mean[d]);
            }
        }
        for (auto& v : cov_means) {
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
        }
    }

    // removeDensitiesWithLowWeight (Mm/MixtureEstimator.cc:64-82), then the first-appearance index maps (:804-817)
    amx_mixture_set* r = new amx_mixture_set;
    r->dim             = dim;
    r->mix_off.assign(1, 0);
    std::unordered_map<uint32_t, uint32_t> dmap, mmap, cmap;
    std::vector<uint32_t>                  dorder, morder, corder;
    std::vector<double>                    lin_weight;
    for (int m = 0; m < n_mix; ++m) {
        std::vector<uint32_t> ed;
        std::vector<double>   ew;
        for (uint32_t k = t->mix_offsets[m]; k < t->mix_offsets[m + 1]; ++k) {
            ed.push_back(t->dens_index[k]);
            ew.push_back(kw[k]);
        }
        if (!ed.empty()) {
            size_t dmax = 0;
            for (size_t i = 1; i < ew.size(); ++i)
                if (ew[i] > ew[dmax])
                    dmax = i;
            double total = 0.0;
            for (double w : ew)
                total += w;
            const double min_w = std::max(c.min_observation_weight, total * c.min_relative_weight);
            for (size_t i = 0; i < ed.size();) {
                if (!(ew[i] >= min_w) && i != dmax) {
                    ed.erase(ed.begin() + (long)i);
                    ew.erase(ew.begin() + (long)i);
                    if (dmax > i)
                        --dmax;
                }
                else
                    ++i;
            }
        }
        for (size_t i = 0; i < ed.size(); ++i) {
            first_seen(mmap, morder, t->dens_mean[ed[i]]);
            first_seen(cmap, corder, t->dens_cov[ed[i]]);
            r->dens_index.push_back(first_seen(dmap, dorder, ed[i]));
            // Mixture::addDensity (Mm/Mixture.cc:63-66)
            r->log_weight.push_back(ew[i] > 0 ? std::log(ew[i]) : -DBL_MAX);
        }
        const size_t a = r->mix_off.back(), b = r->dens_index.size();
        if (c.normalize_mixture_weights && b > a) {  // Mixture::normalizeWeights (:68-74)
            const double norm = log_exp_norm(r->log_weight, a, b);
            for (size_t i = a; i < b; ++i)
                r->log_weight[i] -= norm;
        }
        r->mix_off.push_back((uint32_t)b);
    }
    for (uint32_t d : dorder) {
        r->dens_mean.push_back(mmap[t->dens_mean[d]]);
        r->dens_cov.push_back(cmap[t->dens_cov[d]]);
    }
    // MeanEstimator::estimate (Mm/GaussDensityEstimator.cc:148-158): f64 division, stored as f32; zero weight -> zeros
    std::vector<double> mean_w, cov_w;
    r->means.assign(morder.size() * (size_t)dim, 0.f);
    for (size_t n = 0; n < morder.size(); ++n) {
        const uint32_t o = morder[n];
        mean_w.push_back(mw[o]);
        if (mw[o] != 0)
            for (int i = 0; i < dim; ++i)
                r->means[n * dim + i] = (float)(ms[(size_t)o * dim + i] / mw[o]);
    }
    // CovarianceEstimator::estimate (:194-233): (sum x^2 - sum_j sum_j^2 / N_j) / N, minimum variance; zero weight -> ones
    r->variances.assign(corder.size() * (size_t)dim, 1.f);
    std::vector<double> wmss((size_t)dim);
    const float         min_var = (float)c.min_variance;
    for (size_t n = 0; n < corder.size(); ++n) {
        const uint32_t o = corder[n];
        cov_w.push_back(cw[o]);
        if (cw[o] == 0)
            continue;
        std::fill(wmss.begin(), wmss.end(), 0.0);
        for (uint32_t mi : cov_means[o])
            if (mw[mi] > 0)
                for (int i = 0; i < dim; ++i) {
                    const double y = ms[(size_t)mi * dim + i];
                    wmss[i]        = wmss[i] + y * y / mw[mi];
                }
        for (int i = 0; i < dim; ++i) {
            float v = (float)((cs[(size_t)o * dim + i] - wmss[i]) / cw[o]);
            if (min_var != 0 && v < min_var)
                v = min_var;
            r->variances[n * dim + i] = v;
        }
    }

    if (c.split) {
        // splitMeans (Mm/MixtureSetSplitter.cc:49-65): one pass over the densities -- a mean that several densities share is
        // split once per density, on top of the previous perturbation; the epsilon uses that density's covariance
        const size_t          n_dens0 = r->dens_mean.size(), n_mean0 = morder.size(), n_cov0 = corder.size();
        std::vector<uint32_t> split_mean(n_mean0), split_cov(n_cov0), split_dens(n_dens0);
        std::vector<float>    pert((size_t)dim);
        for (size_t d = 0; d < n_dens0; ++d) {
            const uint32_t mi = r->dens_mean[d], ci = r->dens_cov[d];
            for (int i = 0; i < dim; ++i)
                pert[i] = (float)((double)std::sqrt(r->variances[(size_t)ci * dim + i]) * c.split_perturbation_weight * (double)FLT_EPSILON);
            if (mean_w[mi] > c.split_min_mean_observation_weight) {
                const size_t nm = r->means.size() / (size_t)dim;
                r->means.resize((nm + 1) * (size_t)dim);
                for (int i = 0; i < dim; ++i) {
                    const float v                  = r->means[(size_t)mi * dim + i];
                    r->means[nm * dim + i]         = v - pert[i];
                    r->means[(size_t)mi * dim + i] = v + pert[i];
                }
                split_mean[mi] = (uint32_t)nm;
            }
            else
                split_mean[mi] = mi;
        }
        for (size_t ci = 0; ci < n_cov0; ++ci) {  // splitCovariances (:78-88): a clone
            if (cov_w[ci] > c.split_min_covariance_observation_weight) {
                const size_t nc = r->variances.size() / (size_t)dim;
                r->variances.resize((nc + 1) * (size_t)dim);
                std::copy(r->variances.begin() + (long)(ci * dim), r->variances.begin() + (long)((ci + 1) * dim),
                          r->variances.begin() + (long)(nc * dim));
                split_cov[ci] = (uint32_t)nc;
            }
            else
                split_cov[ci] = (uint32_t)ci;
        }
        for (size_t d = 0; d < n_dens0; ++d) {  // splitDensities (:90-105)
            const uint32_t sm = split_mean[r->dens_mean[d]], sc = split_cov[r->dens_cov[d]];
            if (sm != r->dens_mean[d] || sc != r->dens_cov[d]) {
                split_dens[d] = (uint32_t)r->dens_mean.size();
                r->dens_mean.push_back(sm);
                r->dens_cov.push_back(sc);
            }
            else
                split_dens[d] = (uint32_t)d;
        }
        // splitDensitiesInMixtures (:107-123): the new densities are appended to their mixture with the parent's log weight
        std::vector<uint32_t> off(1, 0), kd;
        std::vector<double>   lw;
        for (int m = 0; m < n_mix; ++m) {
            const size_t a = kd.size();
            for (uint32_t k = r->mix_off[m]; k < r->mix_off[m + 1]; ++k) {
                kd.push_back(r->dens_index[k]);
                lw.push_back(r->log_weight[k]);
            }
            for (uint32_t k = r->mix_off[m]; k < r->mix_off[m + 1]; ++k) {
                const uint32_t sd = split_dens[r->dens_index[k]];
                if (sd != r->dens_index[k]) {
                    kd.push_back(sd);
                    lw.push_back(r->log_weight[k]);
                }
            }
            if (c.split_normalize_mixture_weights && kd.size() > a) {
                const double norm = log_exp_norm(lw, a, kd.size());
                for (size_t i = a; i < kd.size(); ++i)
                    lw[i] -= norm;
            }
            off.push_back((uint32_t)kd.size());
        }
        r->mix_off.swap(off);
        r->dens_index.swap(kd);
        r->log_weight.swap(lw);
    }
    *out = r;
    return AMX_OK;
}

}  // extern "C"
