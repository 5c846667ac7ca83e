// nn_io.cpp -- host-side NN model plumbing: RASR's binary parameter matrices and the state prior.
//
// Parameter files ("bin:<base>-f32-layer-<i>.bin", Nn/NeuralNetwork.cc:542-570) hold one Math::Matrix<f32> per layer
// written by Core::BinaryFormat (Core/FormatSet.hh:245-256) as Math/Matrix.hh:560-563 + Math/Vector.hh:286-290 stream it:
//   u32 nRows, u32 nColumns, u32 nRows, then per row: u32 nColumns, f32 x nColumns      (little endian, no header)
// The matrix is [out x (1 + in)] with column 0 = bias (Nn/LinearLayer.cc:383-420: setParameters).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <string>
#include <vector>

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
    ok          = ok && d != nullptr;
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

extern "C++" {
// ---- Math::Vector<T> files (Math/Module.cc:25-41: "xml" is the default format, "bin:" selects Core::BinaryFormat)
// XML as Core::XmlWriter << Math::Vector<T> writes it (Math/Vector.hh:357-367): declaration, <vector-TYPE size="n">, the
// elements in scientific notation separated by blanks, </vector-TYPE>.  The reader accepts any document whose root element is
// vector-TYPE (optional size attribute checked like Core/VectorParser.hh:85-103: "Vector dimension mismatch").
static bool read_text_file(const char* path, std::string* out) {
    FILE* f = fopen(path, "rb");
    if (!f)
        return false;
    char   buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0)
        out->append(buf, n);
    fclose(f);
    return true;
}

template<class T>
static int vector_read_xml(const char* path, const char* type, int* n, T** data, const char* who) {
    std::string doc;
    if (!read_text_file(path, &doc)) {
        amx::set_error("%s: cannot open '%s'", who, path);
        return AMX_ERR_INVALID;
    }
    // XML comments may quote the element: blank them out before looking for it (Core/VectorParser.hh sits behind a real XML parser)
    for (size_t c0 = doc.find("<!--"); c0 != std::string::npos; c0 = doc.find("<!--", c0)) {
        const size_t c1 = doc.find("-->", c0 + 4);
        const size_t ce = c1 == std::string::npos ? doc.size() : c1 + 3;
        for (size_t i = c0; i < ce; ++i)
            doc[i] = ' ';
    }
    const std::string open_tag = std::string("<vector-") + type;
    size_t            p0 = doc.find(open_tag);
    size_t            p1 = p0 == std::string::npos ? p0 : doc.find('>', p0);
    const size_t      p2 = doc.find(std::string("</vector-") + type);
    if (p0 == std::string::npos || p1 == std::string::npos || p2 == std::string::npos || p2 < p1) {
        amx::set_error("%s: '%s' holds no <vector-%s> element", who, path, type);
        return AMX_ERR_INVALID;
    }
    long              want = -1;
    const std::string head = doc.substr(p0, p1 - p0);
    // size = "3", size='3', size="3": white space around '=' and either quote character, as any XML parser accepts them
    for (size_t ps = head.find("size"); ps != std::string::npos; ps = head.find("size", ps + 4)) {
        if (ps > 0 && !isspace((unsigned char)head[ps - 1]))
            continue;  // part of another attribute name
        size_t q = ps + 4;
        while (q < head.size() && isspace((unsigned char)head[q]))
            ++q;
        if (q >= head.size() || head[q] != '=')
            continue;
        ++q;
        while (q < head.size() && isspace((unsigned char)head[q]))
            ++q;
        if (q < head.size() && (head[q] == '"' || head[q] == '\''))
            ++q;
        if (q < head.size() && isdigit((unsigned char)head[q]))
            want = atol(head.c_str() + q);
        break;
    }
    std::vector<T> v;
    const char*    c = doc.c_str() + p1 + 1;
    const char*    e = doc.c_str() + p2;
    while (c < e) {
        char*  end = nullptr;
        double x   = strtod(c, &end);
        if (end == c) {  // not a number: only white space may remain
            while (c < e && isspace((unsigned char)*c))
                ++c;
            if (c < e) {
                amx::set_error("%s: '%s': unexpected character '%c' in vector data", who, path, *c);
                return AMX_ERR_INVALID;
            }
            break;
        }
        v.push_back((T)x);
        c = end;
    }
    if (want >= 0 && (size_t)want != v.size()) {
        amx::set_error("%s: '%s': Vector dimension mismatch: %ld given and %zu read.", who, path, want, v.size());
        return AMX_ERR_INVALID;
    }
    T* d = (T*)malloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (!d) {
        amx::set_error("%s: out of memory", who);
        return AMX_ERR_INVALID;
    }
    memcpy(d, v.data(), v.size() * sizeof(T));
    *n    = (int)v.size();
    *data = d;
    return AMX_OK;
}

template<class T>
static int vector_write_xml(const char* path, const char* type, int n, const T* data, bool is_float, const char* who) {
    FILE* f = fopen(path, "wb");
    if (!f) {
        amx::set_error("%s: cannot open '%s'", who, path);
        return AMX_ERR_INVALID;
    }
    fprintf(f, "<?xml version=\"1.0\" encoding=\"ISO-8859-1\"?>\n<vector-%s size=\"%d\">\n  ", type, n);
    for (int i = 0; i < n; ++i) {
        if (is_float)
            fprintf(f, "%.20e ", (double)data[i]);  // Prior::write: formats().write(filename, priors, 20)
        else
            fprintf(f, "%ld ", (long)data[i]);
    }
    fprintf(f, "\n</vector-%s>\n", type);
    const bool ok = fclose(f) == 0;
    if (!ok)
        amx::set_error("%s: write to '%s' failed", who, path);
    return ok ? AMX_OK : AMX_ERR_INVALID;
}

}  // extern "C++"

int amx_nn_vector_read_f32(const char* path, int* n, float** data) {
    AMX_REQUIRE(path && n && data, AMX_ERR_INVALID, "amx_nn_vector_read_f32: NULL argument");
    *data = nullptr;
    *n    = 0;
    if (strncmp(path, "bin:", 4) == 0) {  // Math::Vector::read(BinaryInputStream): u32 size, elements
        FILE* f = fopen(path + 4, "rb");
        AMX_REQUIRE(f, AMX_ERR_INVALID, "amx_nn_vector_read_f32: cannot open '%s'", path + 4);
        uint32_t sz = 0;
        bool     ok = read_u32(f, &sz) && sz < (1u << 30);
        float*   d  = ok ? (float*)malloc(std::max<size_t>(sz, 1) * 4) : nullptr;
        ok          = ok && d && fread(d, 4, sz, f) == sz;
        fclose(f);
        if (!ok) {
            free(d);
            amx::set_error("amx_nn_vector_read_f32: '%s' is not a binary Math::Vector<f32>", path + 4);
            return AMX_ERR_INVALID;
        }
        *n    = (int)sz;
        *data = d;
        return AMX_OK;
    }
    const char* p = strncmp(path, "xml:", 4) == 0 ? path + 4 : path;
    return vector_read_xml<float>(p, "f32", n, data, "amx_nn_vector_read_f32");
}

int amx_nn_vector_write_f32(const char* path, int n, const float* data) {
    AMX_REQUIRE(path && n >= 0 && (data || n == 0), AMX_ERR_INVALID, "amx_nn_vector_write_f32: bad argument");
    if (strncmp(path, "bin:", 4) == 0) {
        FILE* f = fopen(path + 4, "wb");
        AMX_REQUIRE(f, AMX_ERR_INVALID, "amx_nn_vector_write_f32: cannot open '%s'", path + 4);
        bool ok = write_u32(f, (uint32_t)n) && fwrite(data, 4, (size_t)n, f) == (size_t)n;
        ok      = (fclose(f) == 0) && ok;
        AMX_REQUIRE(ok, AMX_ERR_INVALID, "amx_nn_vector_write_f32: write to '%s' failed", path + 4);
        return AMX_OK;
    }
    const char* p = strncmp(path, "xml:", 4) == 0 ? path + 4 : path;
    return vector_write_xml<float>(p, "f32", n, data, true, "amx_nn_vector_write_f32");
}

int amx_nn_vector_read_s32(const char* path, int* n, int** data) {
    AMX_REQUIRE(path && n && data, AMX_ERR_INVALID, "amx_nn_vector_read_s32: NULL argument");
    *data = nullptr;
    *n    = 0;
    const char* p = strncmp(path, "xml:", 4) == 0 ? path + 4 : path;
    return vector_read_xml<int>(p, "s32", n, data, "amx_nn_vector_read_s32");
}

int amx_nn_vector_write_s32(const char* path, int n, const int* data) {
    AMX_REQUIRE(path && n >= 0 && (data || n == 0), AMX_ERR_INVALID, "amx_nn_vector_write_s32: bad argument");
    const char* p = strncmp(path, "xml:", 4) == 0 ? path + 4 : path;
    return vector_write_xml<int>(p, "s32", n, data, false, "amx_nn_vector_write_s32");
}

// Nn::ClassLabelWrapper::initMapping (Nn/ClassLabelWrapper.cc:56-70)
int amx_class_labels_init(int n_classes, const int* disregard, int n_disregard, int* mapping, int* n_targets) {
    AMX_REQUIRE(n_classes > 0 && mapping && (disregard || n_disregard == 0), AMX_ERR_INVALID, "amx_class_labels_init: bad argument");
    int next = 0;
    for (int c = 0; c < n_classes; ++c) {
        bool dis = false;
        for (int i = 0; i < n_disregard && !dis; ++i)
            dis = disregard[i] == c;
        mapping[c] = dis ? -1 : next++;
    }
    if (n_targets)
        *n_targets = next;
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
