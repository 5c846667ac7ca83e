// gmm_device.hpp -- device helpers shared by gmm.hip and gmm_fused.hip (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

namespace amx {

// Same arithmetic, two dimensions per instruction: v_pk_add_f32 / v_pk_mul_f32 round each half exactly like the scalar
// operations (nothing is fused), and the partial sums (l0, l1) and (l2, l3) are updated as pairs -- bit-identical to
// gmm_distance at half the instruction count.  mu and is must be 8-byte aligned.
typedef float gmm_pk2 __attribute__((ext_vector_type(2)));

template<int DIM>
__device__ __forceinline__ float gmm_distance_pk(const float (&x)[DIM], const float* __restrict__ mu, const float* __restrict__ is) {
    gmm_pk2       l01 = {0.f, 0.f}, l23 = {0.f, 0.f};
    constexpr int EFF = DIM & ~3;
#pragma unroll
    for (int i = 0; i < EFF; i += 4) {
        const gmm_pk2 d01 = (*(const gmm_pk2*)(mu + i) - gmm_pk2{x[i], x[i + 1]}) * *(const gmm_pk2*)(is + i);
        const gmm_pk2 d23 = (*(const gmm_pk2*)(mu + i + 2) - gmm_pk2{x[i + 2], x[i + 3]}) * *(const gmm_pk2*)(is + i + 2);
        l01               = l01 + d01 * d01;
        l23               = l23 + d23 * d23;
    }
    float result = 0.f;
    result       = result + ((l01.x + l01.y) + (l23.x + l23.y));
#pragma unroll
    for (int i = EFF; i < DIM; ++i) {
        float df = (mu[i] - x[i]) * is[i];
        result   = result + df * df;
    }
    return result;
}

// the same with the mean already in registers (software-pipelined callers)
template<int DIM>
__device__ __forceinline__ float gmm_distance_pk_reg(const float (&x)[DIM], const float (&mu)[DIM], const float* __restrict__ is) {
    gmm_pk2       l01 = {0.f, 0.f}, l23 = {0.f, 0.f};
    constexpr int EFF = DIM & ~3;
#pragma unroll
    for (int i = 0; i < EFF; i += 4) {
        const gmm_pk2 d01 = (gmm_pk2{mu[i], mu[i + 1]} - gmm_pk2{x[i], x[i + 1]}) * *(const gmm_pk2*)(is + i);
        const gmm_pk2 d23 = (gmm_pk2{mu[i + 2], mu[i + 3]} - gmm_pk2{x[i + 2], x[i + 3]}) * *(const gmm_pk2*)(is + i + 2);
        l01               = l01 + d01 * d01;
        l23               = l23 + d23 * d23;
    }
    float result = 0.f;
    result       = result + ((l01.x + l01.y) + (l23.x + l23.y));
#pragma unroll
    for (int i = EFF; i < DIM; ++i) {
        float df = (mu[i] - x[i]) * is[i];
        result   = result + df * df;
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

}  // namespace amx
