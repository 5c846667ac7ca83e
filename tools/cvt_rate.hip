// cvt_rate.hip -- what does v_cvt_scalef32_pk32_fp6_f16 (the q-field conversion of gemm_mx_kernel: 32 f16 -> 32 e2m3 with a block scale)
// cost, alone and beside matrix instructions?  One 512-thread workgroup per CU (two waves per SIMD), per iteration:
//   mode 0: 6 conversions            mode 1: 16 v_mfma_f32_32x32x16_f16 + 8 v_mfma_scale_f32_32x32x64_f8f6f4 (fp6)
//   mode 2: both, conversions first   mode 3: both, the conversions spread between the matrix instructions
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/build/cvt_rate tools/cvt_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef int      v8i __attribute__((ext_vector_type(8)));

__device__ __forceinline__ u32x6 cvt(f16x32 w, float scale) {
    u32x6 q;
    asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2\n\ts_nop 2" : "=&v"(q) : "v"(w), "v"(scale));
    return q;
}

template<int MODE>
__global__ __launch_bounds__(512) void k(int iters, float* sink, float seed) {
    f16x32 w[6];
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 32; ++e)
            w[i][e] = (_Float16)(seed + i + e + threadIdx.x);
    f16x8  a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r)
            acc[i][r] = 0.f;
    unsigned x = 0;
    v8i      av = {1, 2, 3, 4, 5, 6, 0, 0}, bv = {6, 5, 4, 3, 2, 1, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x6 q[6];
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                q[i] = cvt(w[i], 1.0f);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int m = 0; m < 16; ++m)
                acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 7], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 8; ++m)
                acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[m], 2, 2, 0, 127, 0, 127);
        }
        if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 7], 0, 0, 0);
                if (m % 3 == 1 && m / 3 < 6)
                    q[m / 3] = cvt(w[m / 3], 1.0f);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m)
                acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[m], 2, 2, 0, 127, 0, 127);
        }
        if (MODE != 1)
#pragma unroll
            for (int i = 0; i < 6; ++i)
                x ^= q[i][0] ^ q[i][5];
    }
    float s = (float)x;
    for (int i = 0; i < 8; ++i)
        s += acc[i][0];
    if (s == 123.456f)
        sink[threadIdx.x] = s;
}

template<int MODE>
static void run(const char* name, int n_cu, float* sink) {
    const int  iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(n_cu), dim3(512), 0, 0, 1000, sink, 1.f);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(n_cu), dim3(512), 0, 0, iters, sink, 1.f);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("  %-70s %8.1f ns per iteration and SIMD pair of waves\n", name, ms * 1e6 / iters);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    float* sink;
    hipMalloc(&sink, 4096);
    printf("cvt_rate: %d CUs, 8 waves per CU (two per SIMD); an iteration = one K-tile of gemm_mx_kernel per wave\n", prop.multiProcessorCount);
    run<0>("6 x v_cvt_scalef32_pk32_fp6_f16 (+ s_nop 2)", prop.multiProcessorCount, sink);
    run<1>("16 f16 32x32x16 + 8 scaled fp6 32x32x64 matrix instructions", prop.multiProcessorCount, sink);
    run<2>("conversions, then the matrix instructions", prop.multiProcessorCount, sink);
    run<3>("conversions spread between the f16 matrix instructions", prop.multiProcessorCount, sink);
    return 0;
}
