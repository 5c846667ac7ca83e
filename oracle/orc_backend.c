/* oracle/orc_backend.c -- CPU restatement of the feature back-end between the front-end and the scorers
 * (SURVEY.md section 8 row f1).  TEST INFRASTRUCTURE (see orc.h).
 *
 * Signal/Normalization.cc, Regression.cc and MatrixMult.hh sit on Flow::Node / Core::Configuration (boost) and cannot be compiled
 * here as translation units, and the reference has no test vectors for them.  Each function follows the cited source lines
 * operation by operation (types, order of the f32 / f64 operations).  Pinned since round 5: orc_regression on the text of
 * Signal::Regression::regressFirstOrder / regressSecondOrder compiled stand-alone in both flag sets (function-text pin
 * `regression`, oracle/ref/extract_fn.py; tests/test_contract.py), orc_matrix_multiply's row product on Math::Matrix x Vector in
 * both flavours of libref, orc_normalize / orc_normalize_ex (level, mean, mean-and-variance, -1D, divide-by-mean) on the text of
 * Signal::Normalization + Signal::SlidingWindow driven as the node drives them (function-text pin `normalization`: arithmetic and emission
 * order; its sums are contraction-insensitive -- the f64 product of two widened f32 values is exact).  mean-norm: PARITY UNPINNED.
 */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Signal::MeanNormalization / MeanAndVarianceNormalization over one segment (Signal/Normalization.cc:46-66 update,
 * :120-137 mean, :157-187 variance; sliding window Signal/SlidingWindow.hh:401-448).
 * length = right = 0 means "infinite" (whole segment).  For a finite window the frame added at time t evicts frame
 * t - length; frame u leaves while frame u + right is being added, normalised with the running f64 sums at that moment
 * (sum += added, then sum -= removed); the last `right` frames leave at flush time with the statistics of the LAST add
 * (Normalization::update does not touch the statistics while flushing).  type 0 = mean, 1 = mean-and-variance. */
void orc_normalize(const float* in, int n, int dim, int type, int length, int right, float* out) {
    if (n <= 0)
        return;
    const int infinite = (length <= 0);
    double*   sum   = (double*)calloc((size_t)dim, sizeof(double));
    double*   sumsq = (double*)calloc((size_t)dim, sizeof(double));
    float*    mean  = (float*)calloc((size_t)dim, sizeof(float));
    float*    sd    = (float*)calloc((size_t)dim, sizeof(float));
    double    w = 0;
    for (int t = 0; t < n; ++t) {
        const float* x = in + (size_t)t * dim;
        for (int d = 0; d < dim; ++d) {
            sum[d] = sum[d] + (double)x[d];
            sumsq[d] += (double)x[d] * (double)x[d];
        }
        w += 1;
        if (!infinite && t >= length) {
            const float* r = in + (size_t)(t - length) * dim;
            for (int d = 0; d < dim; ++d) {
                sum[d] = sum[d] - (double)r[d];
                sumsq[d] -= (double)r[d] * (double)r[d];
            }
            w -= 1;
        }
        const int emit_now = !infinite && t >= right;
        const int last     = (t == n - 1);
        if (emit_now || last) {
            for (int d = 0; d < dim; ++d) {
                mean[d] = (float)(sum[d] / w);
                if (type == 1) {
                    sd[d] = (float)sqrt((sumsq[d] - sum[d] * sum[d] / w) / w);
                    if (sd[d] == 0)
                        sd[d] = 1.0f;
                }
            }
        }
        if (emit_now) {
            const int    u = t - right;
            const float* s = in + (size_t)u * dim;
            float*       o = out + (size_t)u * dim;
            for (int d = 0; d < dim; ++d) {
                float v = s[d] - mean[d];
                if (type == 1)
                    v = v / sd[d];
                o[d] = v;
            }
        }
    }
    /* flush: frames that have not left yet, with the statistics of the last add */
    const int first = infinite ? 0 : (n - right > 0 ? n - right : 0);
    for (int u = first; u < n; ++u) {
        const float* s = in + (size_t)u * dim;
        float*       o = out + (size_t)u * dim;
        for (int d = 0; d < dim; ++d) {
            float v = s[d] - mean[d];
            if (type == 1)
                v = v / sd[d];
            o[d] = v;
        }
    }
    free(sum);
    free(sumsq);
    free(mean);
    free(sd);
}

/* The other algorithms of signal-normalization (Signal/Normalization.cc:100-110 LevelNormalization, :196-254
 * MeanAndVarianceNormalization1D, :256-262 DivideByMean) on the same sliding-window skeleton (Normalization::update, :47-70):
 *   type 2  divide-by-mean          out = x / mean   (mean as in type 0; the reference stops with "One of the mean components is
 *                                   zero." where this yields inf / NaN)
 *   type 3  level (index `level`)   out[level] = x[level] - max over the window of x[level], the other components pass through
 *   type 4  mean-and-variance-1D    one mean / standard deviation over ALL components of the window's frames:
 *                                   sums run component by component, frame by frame, in f64; sumWeight counts components;
 *                                   mean = (f32)sum / sumWeight (the cast binds to sum first), sd = (f32)sqrt(...), 0 -> 1
 * Statistics follow the window exactly like orc_normalize: added frame first, then the frame that leaves the window. */
void orc_normalize_ex(const float* in, int n, int dim, int type, int level, int length, int right, float* out) {
    if (n <= 0)
        return;
    if (type == 0 || type == 1) {
        orc_normalize(in, n, dim, type, length, right, out);
        return;
    }
    const int infinite = (length <= 0);
    double*   sum  = (double*)calloc((size_t)dim, sizeof(double));
    float*    mean = (float*)calloc((size_t)dim, sizeof(float));
    double    w = 0, sum1 = 0, sumsq1 = 0, w1 = 0;
    float     mean1 = 0, sd1 = 0, mx = 0;
    for (int t = 0; t <= n - 1 + 0; ++t) {
        const float* x = in + (size_t)t * dim;
        for (int d = 0; d < dim; ++d) {
            sum[d] = sum[d] + (double)x[d];
            sumsq1 += (double)x[d] * (double)x[d];
            sum1 += (double)x[d];
        }
        w += 1;
        w1 += dim;
        int lo = 0;  /* oldest frame in the window after this add */
        if (!infinite && t >= length) {
            const float* r = in + (size_t)(t - length) * dim;
            for (int d = 0; d < dim; ++d) {
                sum[d] = sum[d] - (double)r[d];
                sumsq1 -= (double)r[d] * (double)r[d];
                sum1 -= (double)r[d];
            }
            w -= 1;
            w1 -= dim;
        }
        if (!infinite)
            lo = t - length + 1 > 0 ? t - length + 1 : 0;
        const int emit_now = !infinite && t >= right;
        const int last     = (t == n - 1);
        if (emit_now || last) {
            if (type == 2)
                for (int d = 0; d < dim; ++d)
                    mean[d] = (float)(sum[d] / w);
            else if (type == 3) {
                mx = -3.402823466e+38f;  /* Core::Type<f32>::min */
                for (int u = lo; u <= t; ++u)
                    mx = in[(size_t)u * dim + level] > mx ? in[(size_t)u * dim + level] : mx;
            }
            else {
                sd1   = (float)sqrt((sumsq1 - sum1 * sum1 / w1) / w1);
                mean1 = (float)((double)(float)sum1 / w1);
                if (sd1 == 0)
                    sd1 = 1.0f;
            }
        }
        const int u0 = emit_now ? t - right : 0, u1 = emit_now ? t - right + 1 : 0;
        for (int u = u0; u < u1; ++u)
            for (int d = 0; d < dim; ++d) {
                const float v = in[(size_t)u * dim + d];
                out[(size_t)u * dim + d] = type == 2 ? v / mean[d] : type == 3 ? (d == level ? v - mx : v) : (v - mean1) / sd1;
            }
    }
    const int first = infinite ? 0 : (n - right > 0 ? n - right : 0);
    for (int u = first; u < n; ++u)
        for (int d = 0; d < dim; ++d) {
            const float v = in[(size_t)u * dim + d];
            out[(size_t)u * dim + d] = type == 2 ? v / mean[d] : type == 3 ? (d == level ? v - mx : v) : (v - mean1) / sd1;
        }
    free(sum);
    free(mean);
}

/* Signal::Regression::regressFirstOrder / regressSecondOrder (Signal/Regression.cc:25-68) over the window
 * [t - right, t + right] of a segment, missing frames replaced by the closest one (signal-delay, margin-policy copy,
 * margin-condition present-not-empty: Signal/Delay.hh:33-47, derivationWithRegression.flow:7-8). */
void orc_regression(const float* in, int n, int dim, int order, int right, float* out) {
    const int len = 2 * right + 1;
    for (int t = 0; t < n; ++t) {
        float* o = out + (size_t)t * dim;
        for (int c = 0; c < dim; ++c)
            o[c] = 0.0f;
        if (order == 1) {
            float tm = 0.0f;
            for (int i = 0; i < len; ++i) {
                int tt = t - right + i;
                tt     = tt < 0 ? 0 : (tt >= n ? n - 1 : tt);
                const float* f  = in + (size_t)tt * dim;
                const float  dt = (float)((double)(float)i - (double)(float)(len - 1) / 2.0);
                for (int c = 0; c < dim; ++c)
                    o[c] = ORC_FMAF(dt, f[c], o[c]); /* native build: vfmadd213ss */
                tm = ORC_FMAF(dt, dt, tm);           /* vfmadd231ss */
            }
            for (int c = 0; c < dim; ++c)
                o[c] /= tm;
        }
        else {
            float tm = 0.0f, ns = 0.0f;
            for (int i = 0; i < len; ++i) {
                const float dt = (float)((double)(float)i - (double)(float)(len - 1) / 2.0);
                /* dt * dt feeds both sums: a product with a second use that is not an addition stays a product (vmulss, vaddss),
                 * the last product of the fourth power is fused (vfmadd231ss) */
                tm += dt * dt;
                ns = ORC_FMAF(dt * dt * dt, dt, ns);
            }
            ns = ORC_FMAF(tm, tm, -((float)len * ns)); /* vmulss, vfmsub231ss */
            for (int i = 0; i < len; ++i) {
                int tt = t - right + i;
                tt     = tt < 0 ? 0 : (tt >= n ? n - 1 : tt);
                const float* f  = in + (size_t)tt * dim;
                const float  dt = (float)((double)(float)i - (double)(float)(len - 1) / 2.0);
                for (int c = 0; c < dim; ++c) {
                    o[c] = ORC_FMAF(f[c], tm, o[c]);                      /* vfmadd213ss */
                    o[c] = ORC_FMAF(-(f[c] * dt * dt), (float)len, o[c]); /* vmulss, vmulss, vfnmadd213ss */
                }
            }
            for (int c = 0; c < dim; ++c)
                o[c] = (float)((double)o[c] * (2.0 / (double)ns));
        }
    }
}

/* signal-matrix-multiplication-f32: y = M x with Math::Matrix::operator*(Vector) (Math/Matrix.hh:487-494) =
 * one Math::Vector dot product per row, f32 accumulation left to right (Math/Vector.hh:95-101). */
void orc_matrix_multiply(const float* M, int rows, int cols, const float* in, int T, float* out) {
    for (int t = 0; t < T; ++t)
        for (int r = 0; r < rows; ++r) {
            float acc = 0.0f;
            for (int k = 0; k < cols; ++k)
                acc = ORC_FMAF(M[(size_t)r * cols + k], in[(size_t)t * cols + k], acc); /* Math::Vector::operator*, vfmadd231ss */
            out[(size_t)t * rows + r] = acc;
        }
}

/* ------------------------------------------------------------------ signal-vector-f32-*-normalization
 * Signal/VectorNormalization.hh:31-163, one vector at a time: std::inner_product / std::accumulate with a 0.0 (double) seed add
 * the f32-rounded products / the f32 elements to a double in index order; the statistics are then narrowed to f32 (`Value`), and
 * the elements are scaled with f32 operations.  Unqualified sqrt resolves to the double overload (see orc_gammatone.c); narrowing
 * its result to f32 equals sqrtf.  types: 0 amplitude-spectrum-energy, 1 energy, 2 maximum, 3 mean-energy, 4 mean, 5 variance.
 * PINNED on the header's templates taken whole, both builds (oracle/ref/extract_fn.py vector_normalization). */
void orc_vector_normalize(int type, const float* in, int n, int dim, float* out) {
    for (int t = 0; t < n; ++t) {
        const float* v = in + (size_t)t * dim;
        float*       o = out + (size_t)t * dim;
        double       inner = 0.0, acc = 0.0;
        for (int i = 0; i < dim; ++i) {
            float p = v[i] * v[i];
            inner   = inner + p;
            acc     = acc + v[i];
        }
        if (type == 0) { /* amplitude spectrum: first and last bin once, the others twice (Parseval for a real signal) */
            double mid = 0.0;
            for (int i = 1; i < dim - 1; ++i) {
                float p = v[i] * v[i];
                mid     = mid + p;
            }
            float bb   = v[dim - 1] * v[dim - 1];
            float ends = ORC_FMAF(v[0], v[0], bb); /* v.front() * v.front() + v.back() * v.back(): the FIRST product is fused in the default build */
            float sq = (float)sqrt((ends + 2 * mid) / (float)((size_t)(dim - 1) * 2));
            float r  = (float)1 / sq;
            for (int i = 0; i < dim; ++i)
                o[i] = v[i] * r;
        }
        else if (type == 1 || type == 3) {
            float sq = type == 1 ? (float)sqrt(inner) : (float)sqrt(inner / (double)(size_t)dim);
            float r  = (float)1 / sq;
            for (int i = 0; i < dim; ++i)
                o[i] = v[i] * r;
        }
        else if (type == 2) {
            float mx = v[0];
            for (int i = 1; i < dim; ++i)
                if (mx < v[i])
                    mx = v[i]; /* std::max_element: first of the largest */
            float r = (float)1 / mx;
            for (int i = 0; i < dim; ++i)
                o[i] = v[i] * r;
        }
        else if (type == 4) {
            float mean = (float)(acc / (double)(size_t)dim);
            for (int i = 0; i < dim; ++i)
                o[i] = v[i] + -mean;
        }
        else {
            float sum = (float)acc, sumSquare = (float)inner;
            float mean = sum / (float)(size_t)dim;
            float dev  = (float)sqrt((double)((sumSquare - sum * sum / (float)(size_t)dim) / (float)(size_t)dim));
            float r    = (float)1 / dev;
            for (int i = 0; i < dim; ++i) {
                float c = v[i] + -mean;
                o[i]    = c * r;
            }
        }
    }
}

/* generic-vector-f32-<function> (Flow/SimpleFunction.hh:40-345), kinds as AMX_VFUNC_*.  Overloads as the header resolves them: the
 * vector forms of log / ln / exp / sqrt / cos cast std::log10 etc. to T (*)(T) -> the float functions; log-plus, power and quantize
 * call the unqualified log10 / pow / rint on floats -> the double functions, narrowed on assignment (checked for pow with g++ on
 * the reference's headers, see orc_mfcc.c).  PINNED on the header's templates taken whole (oracle/ref/extract_fn.py vector_functions;
 * SimpleFunction.hh itself includes Flow/Node.hh for the node template behind them). */
void orc_vector_function(int kind, float prm, const float* in, long n, int dim, float* out) {
    for (long i = 0; i < n * dim; ++i) {
        float v = in[i], y;
        switch (kind) {
            case 0: y = log10f(v); break;
            case 1: y = (float)log10((double)(v + prm)); break;
            case 2: y = logf(v); break;
            case 3: y = expf(v); break;
            case 4: y = (float)pow((double)v, (double)prm); break;
            case 5: y = sqrtf(v); break;
            case 6: y = cosf(v); break;
            case 7: y = v + prm; break;
            case 8: y = v * prm; break;
            case 9: y = (prm == 1.0f || prm == 0.0f) ? (float)rint((double)v) : (float)(rint((double)(v / prm)) * (double)prm); break;
            case 10: y = fabsf(v); break;
            case 11: y = prm < v ? prm : v; break;
            default: y = v < prm ? prm : v; break;
        }
        out[i] = y;
    }
}
