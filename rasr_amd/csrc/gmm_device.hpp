// gmm_device.hpp -- device helpers shared by gmm.hip and gmm_fused.hip (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

namespace amx {

// The reference's `sum += df * df` in its two builds (amx_gmm_model.tuning contract=off | fma, include/amx.h): built for plain x86-64
// the product and the sum round separately; its default -march=native build on an FMA host contracts them to one vfmadd231ps
// (cmake_resources/CompileOptions.cmake:21-48; tests/test_contract.py: the function text compiled both ways).  This file is compiled with
// -ffp-contract=off, so the fused form only ever comes from here.
template<bool FMA>
__device__ __forceinline__ float sq_acc(float d, float acc) {
    return FMA ? __builtin_fmaf(d, d, acc) : acc + d * d;
}

// Same arithmetic, two dimensions per instruction: v_pk_add_f32 / v_pk_mul_f32 round each half exactly like the scalar
// operations (nothing is fused), and the partial sums (l0, l1) and (l2, l3) are updated as pairs -- bit-identical to
// gmm_distance at half the instruction count.  mu and is must be 8-byte aligned.
typedef float gmm_pk2 __attribute__((ext_vector_type(2)));

template<int DIM, bool FMA = false>
__device__ __forceinline__ float gmm_distance_pk(const float (&x)[DIM], const float* __restrict__ mu, const float* __restrict__ is) {
    gmm_pk2       l01 = {0.f, 0.f}, l23 = {0.f, 0.f};
    constexpr int EFF = DIM & ~3;
#pragma unroll
    for (int i = 0; i < EFF; i += 4) {
        const gmm_pk2 d01 = (*(const gmm_pk2*)(mu + i) - gmm_pk2{x[i], x[i + 1]}) * *(const gmm_pk2*)(is + i);
        const gmm_pk2 d23 = (*(const gmm_pk2*)(mu + i + 2) - gmm_pk2{x[i + 2], x[i + 3]}) * *(const gmm_pk2*)(is + i + 2);
        if (FMA) {
            l01 = gmm_pk2{__builtin_fmaf(d01.x, d01.x, l01.x), __builtin_fmaf(d01.y, d01.y, l01.y)};
            l23 = gmm_pk2{__builtin_fmaf(d23.x, d23.x, l23.x), __builtin_fmaf(d23.y, d23.y, l23.y)};
        }
        else {
            l01 = l01 + d01 * d01;
            l23 = l23 + d23 * d23;
        }
    }
    float result = 0.f;
    result       = result + ((l01.x + l01.y) + (l23.x + l23.y));
#pragma unroll
    for (int i = EFF; i < DIM; ++i) {
        float df = (mu[i] - x[i]) * is[i];
        result   = sq_acc<FMA>(df, result);
    }
    return result;
}

// the same with the mean already in registers (software-pipelined callers)
template<int DIM, bool FMA = false>
__device__ __forceinline__ float gmm_distance_pk_reg(const float (&x)[DIM], const float (&mu)[DIM], const float* __restrict__ is) {
    gmm_pk2       l01 = {0.f, 0.f}, l23 = {0.f, 0.f};
    constexpr int EFF = DIM & ~3;
#pragma unroll
    for (int i = 0; i < EFF; i += 4) {
        const gmm_pk2 d01 = (gmm_pk2{mu[i], mu[i + 1]} - gmm_pk2{x[i], x[i + 1]}) * *(const gmm_pk2*)(is + i);
        const gmm_pk2 d23 = (gmm_pk2{mu[i + 2], mu[i + 3]} - gmm_pk2{x[i + 2], x[i + 3]}) * *(const gmm_pk2*)(is + i + 2);
        if (FMA) {
            l01 = gmm_pk2{__builtin_fmaf(d01.x, d01.x, l01.x), __builtin_fmaf(d01.y, d01.y, l01.y)};
            l23 = gmm_pk2{__builtin_fmaf(d23.x, d23.x, l23.x), __builtin_fmaf(d23.y, d23.y, l23.y)};
        }
        else {
            l01 = l01 + d01 * d01;
            l23 = l23 + d23 * d23;
        }
    }
    float result = 0.f;
    result       = result + ((l01.x + l01.y) + (l23.x + l23.y));
#pragma unroll
    for (int i = EFF; i < DIM; ++i) {
        float df = (mu[i] - x[i]) * is[i];
        result   = sq_acc<FMA>(df, result);
    }
    return result;
}

// IEEE minNum of three without the canonicalising v_max the compiler puts in front of a loop-carried fminf
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float o;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
// The compiler's hazard recognizer does not look into inline assembly: a min3_raw scheduled right behind the MFMA that produces
// its operands reads the accumulator before the last K step has landed.  The FIRST reduction step of every accumulator vector
// therefore goes through the compiler (it inserts the wait states), and the asm steps depend on its result.  (Unnoticed as long
// as the last 16 K columns of the screen operand were zero; dim 48 pooled / dim 24 with per-density covariances put the constant
// columns there and lost every survivor -- found by tools/fuzz_gmm.py.)
__device__ __forceinline__ float min3_first(float a, float b, float c) {
    return __builtin_fminf(__builtin_fminf(a, b), c);
}

// A feature row pre-multiplied by 1 / sigma (Mm::BatchFloatFeatureScorer::setFeature: f * variance_), for the kernels of the
// batch-float family.  DIM > 0: the row lives in registers; DIM == 0 (any other dimension): every element is re-read from global
// memory (L1 / L2 hits) and scaled again -- the same f32 product, so both forms give identical bits.
template<int DIM>
struct ScaledRow {
    float x[DIM > 0 ? DIM : 1];
    __device__ __forceinline__ void load(const float* row, const float* scale, int) {
#pragma unroll
        for (int i = 0; i < DIM; ++i)
            x[i] = row[i] * scale[i];
    }
    __device__ __forceinline__ float operator()(int i) const { return x[i]; }
};
template<>
struct ScaledRow<0> {
    const float* row;
    const float* scale;
    __device__ __forceinline__ void load(const float* r, const float* s, int) {
        row   = r;
        scale = s;
    }
    __device__ __forceinline__ float operator()(int i) const { return row[i] * scale[i]; }
};

// Mm::BatchFloatFeatureScorer's distance (Mm/BatchFeatureScorer.cc:164-254): two 4-lane f32 accumulators over 8-wide blocks, lane 0
// of the first starts at the density's constant; a = s1 + s2; (a3 + a1) + (a2 + a0)
template<int DIM, bool FMA, class Row>
__device__ __forceinline__ float batch_float_distance(const float* __restrict__ mu, const Row& x, float c0, int dim_rt) {
    float s1[4] = {c0, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    auto  block = [&](int d, int n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (d + j < n) {
                float x1 = mu[d + j] - x(d + j);
                s1[j]    = sq_acc<FMA>(x1, s1[j]);  // _mm_add_ps(s1, _mm_mul_ps(x1, x1)): fused in the reference's default build
            }
            if (d + 4 + j < n) {
                float x2 = mu[d + 4 + j] - x(d + 4 + j);
                s2[j]    = sq_acc<FMA>(x2, s2[j]);
            }
        }
    };
    if (DIM > 0) {
#pragma unroll
        for (int d = 0; d < DIM; d += 8)
            block(d, DIM);
    }
    else
        for (int d = 0; d < dim_rt; d += 8)
            block(d, dim_rt);
    const float a0 = s1[0] + s2[0], a1 = s1[1] + s2[1], a2 = s1[2] + s2[2], a3 = s1[3] + s2[3];
    return (a3 + a1) + (a2 + a0);
}

}  // namespace amx
