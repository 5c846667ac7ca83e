// nn_io.cpp -- host-side NN model plumbing: RASR's binary parameter matrices and the state prior.
//
// Parameter files ("bin:<base>-f32-layer-<i>.bin", Nn/NeuralNetwork.cc:542-570) hold one Math::Matrix<f32> per layer
// written by Core::BinaryFormat (Core/FormatSet.hh:245-256) as Math/Matrix.hh:560-563 + Math/Vector.hh:286-290 stream it:
//   u32 nRows, u32 nColumns, u32 nRows, then per row: u32 nColumns, f32 x nColumns      (little endian, no header)
// The matrix is [out x (1 + in)] with column 0 = bias (Nn/LinearLayer.cc:383-420: setParameters).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

namespace {

bool read_u32(FILE* f, uint32_t* v) {
    unsigned char b[4];
    if (fread(b, 1, 4, f) != 4)
        return false;
    *v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    return true;
}

bool write_u32(FILE* f, uint32_t v) {
    unsigned char b[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16), (unsigned char)(v >> 24)};
    return fwrite(b, 1, 4, f) == 4;
}

}  // namespace

extern "C" {

void amx_free(void* p) {
    free(p);
}

int amx_nn_matrix_read(const char* path, int* rows, int* cols, float** data) {
    AMX_REQUIRE(path && rows && cols && data, AMX_ERR_INVALID, "amx_nn_matrix_read: NULL argument");
    *data = nullptr;
    const char* p = strncmp(path, "bin:", 4) == 0 ? path + 4 : path;
    FILE*       f = fopen(p, "rb");
    AMX_REQUIRE(f, AMX_ERR_INVALID, "amx_nn_matrix_read: cannot open '%s'", p);
    uint32_t nr = 0, nc = 0, n2 = 0;
    bool     ok = read_u32(f, &nr) && read_u32(f, &nc) && read_u32(f, &n2) && n2 == nr && (uint64_t)nr * nc < (1ull << 31);
    float*   d  = ok ? (float*)malloc(std::max<size_t>((size_t)nr * nc, 1) * sizeof(float)) : nullptr;
    for (uint32_t r = 0; ok && r < nr; ++r) {
        uint32_t len = 0;
        ok           = read_u32(f, &len) && len == nc && fread(d + (size_t)r * nc, 4, nc, f) == nc;
    }
    fclose(f);
    if (!ok) {
        free(d);
        amx::set_error("amx_nn_matrix_read: '%s' is not a binary Math::Matrix<f32>", p);
        return AMX_ERR_INVALID;
    }
    *rows = (int)nr;
    *cols = (int)nc;
    *data = d;
    return AMX_OK;
}

int amx_nn_matrix_write(const char* path, int rows, int cols, const float* data) {
    AMX_REQUIRE(path && rows >= 0 && cols >= 0 && (data || rows * cols == 0), AMX_ERR_INVALID, "amx_nn_matrix_write: bad argument");
    const char* p = strncmp(path, "bin:", 4) == 0 ? path + 4 : path;
    FILE*       f = fopen(p, "wb");
    AMX_REQUIRE(f, AMX_ERR_INVALID, "amx_nn_matrix_write: cannot open '%s'", p);
    bool ok = write_u32(f, (uint32_t)rows) && write_u32(f, (uint32_t)cols) && write_u32(f, (uint32_t)rows);
    for (int r = 0; ok && r < rows; ++r)
        ok = write_u32(f, (uint32_t)cols) && fwrite(data + (size_t)r * cols, 4, (size_t)cols, f) == (size_t)cols;
    ok = (fclose(f) == 0) && ok;
    AMX_REQUIRE(ok, AMX_ERR_INVALID, "amx_nn_matrix_write: write to '%s' failed", p);
    return AMX_OK;
}

int amx_nn_layer_from_parameters(const float* params, int rows, int cols, int has_bias, float* W, float* bias) {
    AMX_REQUIRE(params && W && rows > 0 && cols > (has_bias ? 1 : 0), AMX_ERR_INVALID, "amx_nn_layer_from_parameters: bad argument");
    const int in = cols - (has_bias ? 1 : 0);
    for (int r = 0; r < rows; ++r) {
        if (bias)
            bias[r] = has_bias ? params[(size_t)r * cols] : 0.f;
        memcpy(W + (size_t)r * in, params + (size_t)r * cols + (has_bias ? 1 : 0), (size_t)in * sizeof(float));
    }
    return AMX_OK;
}

// Nn::Prior<f32>::setFromMixtureSet (Nn/Prior.cc:159-188) with a one-to-one class label mapping: per mixture the f32 sum of
// exp(logWeight) (each term added in f64 and rounded back to f32), normalised by the f32 total, natural log in f32.
int amx_prior_from_mixture_set(const amx_gmm_model* m, float* log_prior) {
    AMX_REQUIRE(m && log_prior && m->mix_offsets && m->log_weight, AMX_ERR_INVALID, "amx_prior_from_mixture_set: NULL argument");
    for (int i = 0; i < m->n_mix; ++i) {
        float acc = 0.f;
        for (uint32_t k = m->mix_offsets[i]; k < m->mix_offsets[i + 1]; ++k)
            acc = (float)((double)acc + std::exp(m->log_weight[k]));
        log_prior[i] = acc;
    }
    double total = 0.0;  // std::accumulate(begin, end, 0.0)
    for (int i = 0; i < m->n_mix; ++i)
        total += (double)log_prior[i];
    const float obs = (float)total;
    for (int i = 0; i < m->n_mix; ++i)
        log_prior[i] = std::log(log_prior[i] / obs);
    return AMX_OK;
}

}  // extern "C"
