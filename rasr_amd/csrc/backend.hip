// backend.hip -- feature back-end between the front-end and the scorers on device-resident
// [frames x dim] matrices: segment-wise mean (/ variance) normalisation, regression derivatives, linear transform (LDA).
// The sliding-window concatenation of the same row is context_window_kernel in mfcc.hip.
//
// Replaces, per Tools/FeatureExtraction/share/processing.standard_system.flow, derivationWithRegression.flow, lda.flow:
//   signal-normalization           Signal/Normalization.cc:46-66,120-187 (+ Signal/SlidingWindow.hh:401-448)
//   signal-delay + signal-regression   Signal/Delay.hh:33-47 (copy margin, present-not-empty), Signal/Regression.cc:25-68
//   signal-matrix-multiplication-f32   Signal/MatrixMult.hh:246-255 -> Math/Matrix.hh:487-494, Math/Vector.hh:95-101
// All three follow the reference's operation order (f64 running sums updated add-then-remove, f32 taps accumulated in
// window order, f32 dot products left to right), so the tests can compare them bit for bit with a CPU restatement.
#include "common.hpp"

#include <algorithm>

// internal view of an MFCC plan's segmentation (mfcc.hip)
extern "C" int amx_internal_plan_view(const amx_mfcc_plan* p, const long long** d_frame_off, int* n_seg, long long* total);

namespace amx {

// One workgroup per segment, one lane per dimension (strided), frames in order: the reference's statistics are a
// sequential recurrence per dimension.  64 segments x 40 dimensions of work are tiny next to the scorers.
__global__ __launch_bounds__(64) void normalize_kernel(const float* __restrict__ in, int in_ld, const long long* __restrict__ frame_off,
                                                      int dim, int type, int length, int right, float* __restrict__ out, int out_ld) {
    const long long s0 = frame_off[blockIdx.x], s1 = frame_off[blockIdx.x + 1];
    const int       n  = (int)(s1 - s0);
    if (n <= 0)
        return;
    const bool infinite = length <= 0;
    for (int d = threadIdx.x; d < dim; d += 64) {
        double sum = 0.0, sumsq = 0.0, w = 0.0;
        float  mean = 0.f, sd = 1.f;
        for (int t = 0; t < n; ++t) {
            const double x = (double)in[(s0 + t) * in_ld + d];
            sum            = sum + x;
            sumsq += x * x;
            w += 1.0;
            if (!infinite && t >= length) {
                const double r = (double)in[(s0 + t - length) * in_ld + d];
                sum            = sum - r;
                sumsq -= r * r;
                w -= 1.0;
            }
            const bool emit_now = !infinite && t >= right;
            if (emit_now || t == n - 1) {
                mean = (float)(sum / w);
                if (type == 1) {
                    sd = (float)sqrt((sumsq - sum * sum / w) / w);
                    if (sd == 0.f)
                        sd = 1.f;
                }
            }
            if (emit_now) {
                float v = in[(s0 + t - right) * in_ld + d] - mean;
                if (type == 1)
                    v = v / sd;
                out[(s0 + t - right) * out_ld + d] = v;
            }
        }
        const int first = infinite ? 0 : max(n - right, 0);
        for (int u = first; u < n; ++u) {  // flush: statistics of the last add
            float v = in[(s0 + u) * in_ld + d] - mean;
            if (type == 1)
                v = v / sd;
            out[(s0 + u) * out_ld + d] = v;
        }
    }
}

// divide-by-mean (type 2), level (3) and mean-and-variance-1D (4): Signal/Normalization.cc:100-110,196-262.  One workgroup per
// segment.  The 1D variant's statistics are ONE f64 recurrence over all components of all frames in the reference's order
// (component by component, frame by frame), so lane 0 carries them; the window maximum of the level variant is rescanned per
// emitted frame like LevelNormalization::finalize does; the per-frame results go through LDS to the lanes that apply them.
__global__ __launch_bounds__(64) void normalize_misc_kernel(const float* __restrict__ in, int in_ld, const long long* __restrict__ frame_off,
                                                           int dim, int type, int level, int length, int right, float* __restrict__ out,
                                                           int out_ld) {
    const long long s0 = frame_off[blockIdx.x], s1 = frame_off[blockIdx.x + 1];
    const int       n  = (int)(s1 - s0);
    if (n <= 0)
        return;
    const bool infinite = length <= 0;
    const int  lane = threadIdx.x;
    if (type == 2) {  // per dimension, like normalize_kernel
        for (int d = lane; d < dim; d += 64) {
            double sum = 0.0, w = 0.0;
            float  mean = 0.f;
            for (int t = 0; t < n; ++t) {
                sum = sum + (double)in[(s0 + t) * in_ld + d];
                w += 1.0;
                if (!infinite && t >= length) {
                    sum = sum - (double)in[(s0 + t - length) * in_ld + d];
                    w -= 1.0;
                }
                const bool emit_now = !infinite && t >= right;
                if (emit_now || t == n - 1)
                    mean = (float)(sum / w);
                if (emit_now)
                    out[(s0 + t - right) * out_ld + d] = in[(s0 + t - right) * in_ld + d] / mean;
            }
            for (int u = infinite ? 0 : max(n - right, 0); u < n; ++u)
                out[(s0 + u) * out_ld + d] = in[(s0 + u) * in_ld + d] / mean;
        }
        return;
    }
    __shared__ float s_a, s_b;  // type 3: window maximum;  type 4: mean, standard deviation
    double sum1 = 0.0, sumsq1 = 0.0, w1 = 0.0;
    auto   apply = [&](int u) {
        const float a = s_a, b = s_b;
        for (int d = lane; d < dim; d += 64) {
            const float v = in[(s0 + u) * in_ld + d];
            out[(s0 + u) * out_ld + d] = type == 3 ? (d == level ? v - a : v) : (v - a) / b;
        }
    };
    for (int t = 0; t < n; ++t) {
        const bool emit_now = !infinite && t >= right;
        const bool refresh  = emit_now || t == n - 1;
        if (lane == 0) {
            if (type == 4) {
                for (int d = 0; d < dim; ++d) {
                    const double x = (double)in[(s0 + t) * in_ld + d];
                    sumsq1 += x * x;
                    sum1 += x;
                }
                w1 += dim;
                if (!infinite && t >= length) {
                    for (int d = 0; d < dim; ++d) {
                        const double r = (double)in[(s0 + t - length) * in_ld + d];
                        sumsq1 -= r * r;
                        sum1 -= r;
                    }
                    w1 -= dim;
                }
                if (refresh) {
                    float sd = (float)sqrt((sumsq1 - sum1 * sum1 / w1) / w1);
                    s_a      = (float)((double)(float)sum1 / w1);
                    s_b      = sd == 0.f ? 1.f : sd;
                }
            }
            else if (refresh) {
                const int lo = infinite ? 0 : max(t - length + 1, 0);
                float     mx = -3.402823466e+38f;
                for (int u = lo; u <= t; ++u)
                    mx = fmaxf(in[(s0 + u) * in_ld + level], mx);
                s_a = mx;
                s_b = 1.f;
            }
        }
        __syncthreads();
        if (emit_now)
            apply(t - right);
        __syncthreads();
    }
    for (int u = infinite ? 0 : max(n - right, 0); u < n; ++u)
        apply(u);
}

__device__ __forceinline__ int segment_of(const long long* __restrict__ frame_off, int n_seg, long long t) {
    int lo = 0, hi = n_seg;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frame_off[mid] <= t)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// one workgroup per frame, lanes over the coefficients
template<bool FMA>   // FMA: the reference's default build (Signal/Regression.cc:24-65 compiled -march=native; INTEGRATION.md lists the contracted sites)
__global__ __launch_bounds__(64) void regression_kernel(const float* __restrict__ in, int in_ld, const long long* __restrict__ frame_off,
                                                       int n_seg, int dim, int order, int right, float* __restrict__ out, int out_ld) {
    const long long t   = blockIdx.x;
    const int       seg = segment_of(frame_off, n_seg, t);
    const long long s0 = frame_off[seg], s1 = frame_off[seg + 1];
    const int       len = 2 * right + 1;
    for (int c = threadIdx.x; c < dim; c += 64) {
        float o = 0.f;
        if (order == 1) {
            float tm = 0.f;
            for (int i = 0; i < len; ++i) {
                long long tt = t - right + i;
                tt           = tt < s0 ? s0 : (tt >= s1 ? s1 - 1 : tt);
                const float dt = (float)((double)(float)i - (double)(float)(len - 1) / 2.0);
                o              = mad<FMA>(dt, in[tt * in_ld + c], o);   // vfmadd213ss
                tm             = mad<FMA>(dt, dt, tm);                  // vfmadd231ss
            }
            o = o / tm;
        }
        else {
            float tm = 0.f, ns = 0.f;
            for (int i = 0; i < len; ++i) {
                const float dt = (float)((double)(float)i - (double)(float)(len - 1) / 2.0);
                tm             = tm + dt * dt;                     // dt * dt feeds both sums: stays a product (vmulss, vaddss)
                ns             = mad<FMA>(dt * dt * dt, dt, ns);   // the last product of the fourth power is fused
            }
            ns = mad<FMA>(tm, tm, -((float)len * ns));             // vmulss, vfmsub231ss
            for (int i = 0; i < len; ++i) {
                long long tt = t - right + i;
                tt           = tt < s0 ? s0 : (tt >= s1 ? s1 - 1 : tt);
                const float f  = in[tt * in_ld + c];
                const float dt = (float)((double)(float)i - (double)(float)(len - 1) / 2.0);
                o              = mad<FMA>(f, tm, o);                          // vfmadd213ss
                o              = mad<FMA>(-(f * dt * dt), (float)len, o);     // vmulss, vmulss, vfnmadd213ss
            }
            o = (float)((double)o * (2.0 / (double)ns));
        }
        out[t * out_ld + c] = o;
    }
}

// y[t][r] = sum_k M[r][k] x[t][k], f32, k ascending.  lane = frame (its row streams through L1), matrix rows are
// wave-uniform -> scalar loads.  A 45 x 440 LDA over 64 k frames is 2.5 GFLOP: not worth an MFMA path that would change the sums.
template<bool FMA>   // FMA: Math::Vector::operator* of the default build (result += a[i] * b[i] is one vfmadd231ss)
__global__ __launch_bounds__(256) void matrix_multiply_kernel(const float* __restrict__ M, int rows, int cols, const float* __restrict__ in,
                                                             int in_ld, int T, float* __restrict__ out, int out_ld) {
    const int t  = blockIdx.x * 256 + threadIdx.x;
    const int tt = t < T ? t : T - 1;
    const float* x = in + (size_t)tt * in_ld;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const float* m   = M + (size_t)r * cols;
        float        acc = 0.f;
        for (int k = 0; k < cols; ++k)
            acc = mad<FMA>(m[k], x[k], acc);
        if (t < T)
            out[(size_t)t * out_ld + r] = acc;
    }
}

}  // namespace amx

// generic-vector-f32-{log, log-plus, ln, exp, power, sqrt, cos, addition, multiplication, quantize, abs, minimum, maximum}
// (Flow/SimpleFunction.hh:40-345): one element per thread.  The arithmetic ones are exact; log / ln / exp / cos are the device's f32
// functions (the reference calls std::log10 / std::log / std::exp / std::cos on floats: glibc's, a few ulp apart); power is the
// unqualified pow on floats = ::pow(double, double) narrowed (see power_node in mfcc.hip), quantize is rint(v / p) * p with the
// unqualified rint on a float = ::rint(double).
__global__ __launch_bounds__(256) void vector_function_kernel(const float* __restrict__ in, int in_ld, long long n, int dim, int kind, float prm,
                                                             float* __restrict__ out, int out_ld) {
    const long long total = n * dim;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / dim;
        const int       c = (int)(i - r * dim);
        const float     v = in[r * in_ld + c];
        float           y;
        switch (kind) {
            case AMX_VFUNC_LOG: y = log10f(v); break;
            case AMX_VFUNC_LOG_PLUS: y = log10f(v + prm); break;
            case AMX_VFUNC_LN: y = logf(v); break;
            case AMX_VFUNC_EXP: y = expf(v); break;
            case AMX_VFUNC_POWER: y = (float)pow((double)v, (double)prm); break;
            case AMX_VFUNC_SQRT: y = sqrtf(v); break;
            case AMX_VFUNC_COS: y = cosf(v); break;
            case AMX_VFUNC_ADDITION: y = v + prm; break;
            case AMX_VFUNC_MULTIPLICATION: y = v * prm; break;
            case AMX_VFUNC_QUANTIZE:
                y = (prm == 1.0f || prm == 0.0f) ? (float)rint((double)v) : (float)(rint((double)(v / prm)) * (double)prm);
                break;
            case AMX_VFUNC_ABS: y = fabsf(v); break;
            case AMX_VFUNC_MINIMUM: y = prm < v ? prm : v; break;   // std::min(a, value): value only if value < a
            default: y = v < prm ? prm : v; break;                  // std::max(a, value): value only if a < value
        }
        out[r * out_ld + c] = y;
    }
}

namespace {
// Do the strided views in [T x in_w] (row stride in_ld) and out [T x out_w] (row stride out_ld) share memory?  Views into one wide
// matrix (same stride) are disjoint when their column ranges are; anything else that overlaps in address range counts as aliasing.
// signal-vector-f32-*-normalization (Signal/VectorNormalization.hh:31-163): one vector at a time, thread = vector.  The statistics
// are std::inner_product / std::accumulate with a double seed -- f32-rounded products (or the elements) added to a double in
// index order --, narrowed to f32; the elements are scaled with f32 operations.  A thread reads its whole row before it writes
// the first element, so the identical view may be normalised in place.
__global__ __launch_bounds__(256) void vector_normalize_kernel(const float* __restrict__ in, int in_ld, long long n, int dim, int type,
                                                              float* __restrict__ out, int out_ld, int fma) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n)
        return;
    const float* v = in + t * in_ld;
    float*       o = out + t * out_ld;
    double       inner = 0.0, acc = 0.0, mid = 0.0;
    float        mx = v[0];
    for (int i = 0; i < dim; ++i) {
        const float x = v[i], p = x * x;
        inner         = inner + (double)p;
        acc           = acc + (double)x;
        if (i > 0 && i < dim - 1)
            mid = mid + (double)p;
        mx = (i > 0 && mx < x) ? x : mx;
    }
    float sub = 0.f, r = 1.f;
    if (type == AMX_VNORM_AMPLITUDE_SPECTRUM_ENERGY) {
        const float ff = v[0] * v[0], bb = v[dim - 1] * v[dim - 1];
        // v.front() * v.front() + v.back() * v.back(): the FIRST product is fused in the reference's default build
        const float ends = fma ? __builtin_fmaf(v[0], v[0], bb) : ff + bb;
        r = (float)1 / (float)sqrt(((double)ends + 2 * mid) / (double)(float)((size_t)(dim - 1) * 2));
    }
    else if (type == AMX_VNORM_ENERGY)
        r = (float)1 / (float)sqrt(inner);
    else if (type == AMX_VNORM_MEAN_ENERGY)
        r = (float)1 / (float)sqrt(inner / (double)(size_t)dim);
    else if (type == AMX_VNORM_MAXIMUM)
        r = (float)1 / mx;
    else if (type == AMX_VNORM_MEAN)
        sub = (float)(acc / (double)(size_t)dim);
    else {  // variance
        const float sum = (float)acc, sumSquare = (float)inner, fn = (float)(size_t)dim;
        sub             = sum / fn;
        const float q   = sum * sum;
        const float e   = (sumSquare - q / fn) / fn;
        r               = (float)1 / (float)sqrt((double)e);
    }
    if (type == AMX_VNORM_MEAN) {
        for (int i = 0; i < dim; ++i)
            o[i] = v[i] + -sub;
    }
    else if (type == AMX_VNORM_VARIANCE) {
        for (int i = 0; i < dim; ++i) {
            const float c = v[i] + -sub;
            o[i]          = c * r;
        }
    }
    else
        for (int i = 0; i < dim; ++i)
            o[i] = v[i] * r;
}

bool views_alias(const float* in, int in_ld, int in_w, const float* out, int out_ld, int out_w, long long T) {
    if (T <= 0)
        return false;
    const float* in_end  = in + (T - 1) * (long long)in_ld + in_w;
    const float* out_end = out + (T - 1) * (long long)out_ld + out_w;
    if (in_end <= out || out_end <= in)
        return false;
    if (in_ld == out_ld) {
        const long long ld = in_ld, delta = out - in;
        const long long c  = ((delta % ld) + ld) % ld;  // column offset of `out` relative to `in`
        if (c >= in_w && c + out_w <= ld)
            return false;  // disjoint column ranges of the same matrix (also with a row shift)
    }
    return true;
}
}  // namespace

extern "C" {

int amx_normalize_dev(amx_ctx* ctx, const amx_mfcc_plan* plan, const float* in_dev, int in_ld, int dim, int type, int length, int right,
                      float* out_dev, int out_ld) {
    AMX_REQUIRE(ctx && plan && in_dev && out_dev, AMX_ERR_INVALID, "amx_normalize_dev: NULL argument");
    AMX_REQUIRE(dim > 0 && in_ld >= dim && out_ld >= dim, AMX_ERR_INVALID, "amx_normalize_dev: bad dimension / stride");
    AMX_REQUIRE(type == AMX_NORM_MEAN || type == AMX_NORM_MEAN_AND_VARIANCE, AMX_ERR_INVALID, "amx_normalize_dev: unknown type %d", type);
    // SlidingWindow::init: "return false if maxSize <= right" -> the node reports "Cannot initialize with parameters ..."
    AMX_REQUIRE(length == 0 || (length > 0 && right >= 0 && right < length), AMX_ERR_INVALID,
                "amx_normalize_dev: Cannot initialize with parameters length (%d), right (%d)", length, right);
    const long long* d_off;
    int              n_seg;
    long long        total;
    int              r = amx_internal_plan_view(plan, &d_off, &n_seg, &total);
    if (r != AMX_OK || total == 0)
        return r;
    // the sliding window re-reads in[t - length] after out[t - right] has been written: in place is only defined for whole
    // segments with identical views (every frame is read before the first one is written back)
    const bool same_view = in_dev == out_dev && in_ld == out_ld;
    AMX_REQUIRE(!views_alias(in_dev, in_ld, dim, out_dev, out_ld, dim, total) || (same_view && length == 0), AMX_ERR_INVALID,
                "amx_normalize_dev: input and output views overlap (in place is only supported for whole-segment normalisation)");
    AMX_HIP(hipSetDevice(ctx->device));
    amx::ScopedKernelTimer timer(ctx, "normalize");
    hipLaunchKernelGGL(amx::normalize_kernel, dim3(n_seg), dim3(64), 0, ctx->stream, in_dev, in_ld, d_off, dim, type, length, right, out_dev, out_ld);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_normalize_ex_dev(amx_ctx* ctx, const amx_mfcc_plan* plan, const float* in_dev, int in_ld, int dim, int type, int level, int length,
                         int right, float* out_dev, int out_ld) {
    if (type == AMX_NORM_MEAN || type == AMX_NORM_MEAN_AND_VARIANCE)
        return amx_normalize_dev(ctx, plan, in_dev, in_ld, dim, type, length, right, out_dev, out_ld);
    AMX_REQUIRE(ctx && plan && in_dev && out_dev, AMX_ERR_INVALID, "amx_normalize_ex_dev: NULL argument");
    AMX_REQUIRE(dim > 0 && in_ld >= dim && out_ld >= dim, AMX_ERR_INVALID, "amx_normalize_ex_dev: bad dimension / stride");
    AMX_REQUIRE(type == AMX_NORM_DIVIDE_BY_MEAN || type == AMX_NORM_LEVEL || type == AMX_NORM_MEAN_AND_VARIANCE_1D, AMX_ERR_INVALID,
                "amx_normalize_ex_dev: unknown type %d", type);
    AMX_REQUIRE(type != AMX_NORM_LEVEL || (level >= 0 && level < dim), AMX_ERR_INVALID, "amx_normalize_ex_dev: level index %d outside the vector", level);
    AMX_REQUIRE(length == 0 || (length > 0 && right >= 0 && right < length), AMX_ERR_INVALID,
                "amx_normalize_ex_dev: Cannot initialize with parameters length (%d), right (%d)", length, right);
    const long long* d_off;
    int              n_seg;
    long long        total;
    int              r = amx_internal_plan_view(plan, &d_off, &n_seg, &total);
    if (r != AMX_OK || total == 0)
        return r;
    const bool same_view = in_dev == out_dev && in_ld == out_ld;
    AMX_REQUIRE(!views_alias(in_dev, in_ld, dim, out_dev, out_ld, dim, total) || (same_view && length == 0), AMX_ERR_INVALID,
                "amx_normalize_ex_dev: input and output views overlap (in place is only supported for whole-segment normalisation)");
    AMX_HIP(hipSetDevice(ctx->device));
    amx::ScopedKernelTimer timer(ctx, "normalize");
    hipLaunchKernelGGL(amx::normalize_misc_kernel, dim3(n_seg), dim3(64), 0, ctx->stream, in_dev, in_ld, d_off, dim, type, level, length, right,
                       out_dev, out_ld);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_vector_normalize_dev(amx_ctx* ctx, int type, const float* in_dev, int in_ld, long n_vectors, int dim, float* out_dev, int out_ld) {
    AMX_REQUIRE(ctx && (n_vectors == 0 || (in_dev && out_dev)), AMX_ERR_INVALID, "amx_vector_normalize_dev: NULL argument");
    AMX_REQUIRE(type >= AMX_VNORM_AMPLITUDE_SPECTRUM_ENERGY && type <= AMX_VNORM_VARIANCE, AMX_ERR_INVALID,
                "amx_vector_normalize_dev: unknown type %d", type);
    AMX_REQUIRE(n_vectors >= 0 && dim > 0 && in_ld >= dim && out_ld >= dim, AMX_ERR_INVALID, "amx_vector_normalize_dev: bad shape / stride");
    AMX_REQUIRE(type != AMX_VNORM_AMPLITUDE_SPECTRUM_ENERGY || dim >= 2, AMX_ERR_INVALID,
                "amx_vector_normalize_dev: an amplitude spectrum has at least two bins");
    if (n_vectors == 0)
        return AMX_OK;
    const bool same_view = in_dev == out_dev && in_ld == out_ld;
    AMX_REQUIRE(same_view || !views_alias(in_dev, in_ld, dim, out_dev, out_ld, dim, n_vectors), AMX_ERR_INVALID,
                "amx_vector_normalize_dev: input and output views overlap (in place is supported on the identical view only)");
    AMX_HIP(hipSetDevice(ctx->device));
    amx::ScopedKernelTimer timer(ctx, "normalize");
    hipLaunchKernelGGL(vector_normalize_kernel, dim3((unsigned)((n_vectors + 255) / 256)), dim3(256), 0, ctx->stream, in_dev, in_ld,
                       (long long)n_vectors, dim, type, out_dev, out_ld, ctx->contract == AMX_CONTRACT_FMA ? 1 : 0);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_vector_function_dev(amx_ctx* ctx, int kind, float parameter, const float* in_dev, int in_ld, long n_vectors, int dim, float* out_dev,
                            int out_ld) {
    AMX_REQUIRE(ctx && in_dev && out_dev, AMX_ERR_INVALID, "amx_vector_function_dev: NULL argument");
    AMX_REQUIRE(kind >= AMX_VFUNC_LOG && kind <= AMX_VFUNC_MAXIMUM, AMX_ERR_INVALID, "amx_vector_function_dev: unknown function %d", kind);
    AMX_REQUIRE(n_vectors >= 0 && dim > 0 && in_ld >= dim && out_ld >= dim, AMX_ERR_INVALID, "amx_vector_function_dev: bad shape / stride");
    if (n_vectors == 0)
        return AMX_OK;
    const bool same_view = in_dev == out_dev && in_ld == out_ld;
    AMX_REQUIRE(same_view || !views_alias(in_dev, in_ld, dim, out_dev, out_ld, dim, n_vectors), AMX_ERR_INVALID,
                "amx_vector_function_dev: input and output views overlap (in place is supported on the identical view only)");
    AMX_HIP(hipSetDevice(ctx->device));
    amx::ScopedKernelTimer timer(ctx, "normalize");
    const long long total = (long long)n_vectors * dim;
    hipLaunchKernelGGL(vector_function_kernel, dim3((unsigned)std::min<long long>(16384, (total + 255) / 256)), dim3(256), 0, ctx->stream, in_dev,
                       in_ld, (long long)n_vectors, dim, kind, parameter, out_dev, out_ld);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_regression_dev(amx_ctx* ctx, const amx_mfcc_plan* plan, const float* in_dev, int in_ld, int dim, int order, int right,
                       float* out_dev, int out_ld) {
    AMX_REQUIRE(ctx && plan && in_dev && out_dev, AMX_ERR_INVALID, "amx_regression_dev: NULL argument");
    AMX_REQUIRE(dim > 0 && in_ld >= dim && out_ld >= dim && right >= 1, AMX_ERR_INVALID, "amx_regression_dev: bad dimension / stride / window");
    // RegressionNode::merge: criticalError("signal-regression only implemented for 1st and 2nd order derivatives")
    AMX_REQUIRE(order == 1 || order == 2, AMX_ERR_UNSUPPORTED, "signal-regression only implemented for 1st and 2nd order derivatives");
    const long long* d_off;
    int              n_seg;
    long long        total;
    int              r = amx_internal_plan_view(plan, &d_off, &n_seg, &total);
    if (r != AMX_OK || total == 0)
        return r;
    // one workgroup per frame reads its neighbours' rows while they write theirs
    AMX_REQUIRE(!views_alias(in_dev, in_ld, dim, out_dev, out_ld, dim, total), AMX_ERR_INVALID,
                "amx_regression_dev: input and output views overlap");
    AMX_HIP(hipSetDevice(ctx->device));
    amx::ScopedKernelTimer timer(ctx, "regression");
    hipLaunchKernelGGL(ctx->contract == AMX_CONTRACT_FMA ? amx::regression_kernel<true> : amx::regression_kernel<false>, dim3((unsigned)total),
                       dim3(64), 0, ctx->stream, in_dev, in_ld, d_off, n_seg, dim, order, right, out_dev, out_ld);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_matrix_multiply_dev(amx_ctx* ctx, const float* matrix_dev, int rows, int cols, const float* in_dev, int in_ld, int T, float* out_dev,
                            int out_ld) {
    AMX_REQUIRE(ctx && matrix_dev && in_dev && out_dev, AMX_ERR_INVALID, "amx_matrix_multiply_dev: NULL argument");
    // MatrixMultiplicationNode::work: "vector/matrix dimension mismatch" when the input size differs from nColumns
    AMX_REQUIRE(rows > 0 && cols > 0 && in_ld >= cols && out_ld >= rows && T >= 0, AMX_ERR_INVALID,
                "amx_matrix_multiply_dev: vector/matrix dimension mismatch: vector stride %d, matrix %d columns", in_ld, cols);
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(!views_alias(in_dev, in_ld, cols, out_dev, out_ld, rows, T), AMX_ERR_INVALID,
                "amx_matrix_multiply_dev: input and output views overlap");
    AMX_HIP(hipSetDevice(ctx->device));
    amx::ScopedKernelTimer timer(ctx, "matrix_multiply");
    hipLaunchKernelGGL(ctx->contract == AMX_CONTRACT_FMA ? amx::matrix_multiply_kernel<true> : amx::matrix_multiply_kernel<false>,
                       dim3((T + 255) / 256, std::min(rows, 64)), dim3(256), 0, ctx->stream, matrix_dev, rows, cols, in_dev, in_ld, T, out_dev, out_ld);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

}  // extern "C"
