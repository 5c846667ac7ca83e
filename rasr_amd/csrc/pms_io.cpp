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
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>

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
    for (unsigned m = 0; m < nMix; ++m) {
        unsigned k = 0;
        in >> k;
        for (unsigned j = 0; j < k; ++j) {
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
    for (unsigned d = 0; d < nDns; ++d) {
        unsigned mi = 0, ci = 0;
        in >> mi >> ci;
        ms->dens_mean.push_back(mi);
        ms->dens_cov.push_back(ci);
    }
    for (unsigned i = 0; i < nMean; ++i) {
        unsigned n = 0;
        in >> n;
        if (n != dim) {
            amx::set_error("amx_pms_read: mean %u has dimension %u, expected %u", i, n, dim);
            delete ms;
            return AMX_ERR_INVALID;
        }
        for (unsigned j = 0; j < n; ++j) {
            float v = 0;
            in >> v;
            ms->means.push_back(v);
        }
    }
    for (unsigned i = 0; i < nCov; ++i) {
        unsigned n = 0;
        in >> n;
        if (n != dim) {
            amx::set_error("amx_pms_read: covariance %u has dimension %u, expected %u", i, n, dim);
            delete ms;
            return AMX_ERR_INVALID;
        }
        for (unsigned j = 0; j < n; ++j) {
            float  v = 0;
            double w = 0;
            in >> v >> w;
            ms->variances.push_back((float)(v * w));
        }
    }
    if (in.fail()) {
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
    v->mixture_weight_scale = 1.f;
    v->gaussian_scale       = 1.f;
    return AMX_OK;
}

void amx_mixture_set_destroy(amx_mixture_set* ms) {
    delete ms;
}

}  // extern "C"
