// tools/ceilings.hip -- measured ceilings of the box next to the datasheet numbers (SURVEY.md section 8d asks for them):
// HBM copy / triad bandwidth, bf16 and f16 MFMA rate from registers (zero and random operands: the chip is power managed),
// f32 FMA rate (plain and packed).  Prints one JSON object.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ceilings.hip -o tools/build/ceilings && tools/build/ceilings
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

typedef __attribute__((ext_vector_type(8))) short    bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float   f32x16;
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void triad(f32x4* __restrict__ a, const f32x4* __restrict__ b, const f32x4* __restrict__ c, size_t n, float s) {
    // one block per 4 x 256 float4s, four loads per operand in flight per lane
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    f32x4        x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        x[u] = __builtin_nontemporal_load(b + base + u * 256);
        y[u] = __builtin_nontemporal_load(c + base + u * 256);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        __builtin_nontemporal_store(x[u] + s * y[u], a + base + u * 256);
}

__global__ __launch_bounds__(256) void fill4(f32x4* __restrict__ a, float v) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const f32x4  x    = {v, v, v, v};
#pragma unroll
    for (int u = 0; u < 4; ++u)
        a[base + u * 256] = x;
}

__global__ __launch_bounds__(256) void read4(const f32x4* __restrict__ b, float* __restrict__ sink) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    f32x4        acc  = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u)
        acc += __builtin_nontemporal_load(b + base + u * 256);
    if (acc.x + acc.y + acc.z + acc.w == 123.456f)
        sink[threadIdx.x] = acc.x;
}

// 64-byte pieces at a 40 KB row stride, the store pattern of a [frames x 10000] f32 score matrix written 16 mixtures at a time
__global__ __launch_bounds__(256) void fill_pieces(f32x4* __restrict__ a, float v, int n_rows, int row_f4, int pieces_per_row) {
    const int    tid = threadIdx.x;
    const f32x4  x   = {v, v, v, v};
    const int    piece = blockIdx.x % pieces_per_row, rb = blockIdx.x / pieces_per_row;  // 256 rows per block
    for (int it = 0; it < 4; ++it) {
        const int e = it * 256 + tid, row = rb * 256 + (e >> 2), c = e & 3;
        if (row < n_rows)
            a[(size_t)row * row_f4 + piece * 4 + c] = x;
    }
}

__global__ __launch_bounds__(256) void copy4(f32x4* __restrict__ a, const f32x4* __restrict__ b) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    f32x4        x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        x[u] = __builtin_nontemporal_load(b + base + u * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        __builtin_nontemporal_store(x[u], a + base + u * 256);
}

template<bool F16>
__global__ __launch_bounds__(256) void mfma_rate(const unsigned* __restrict__ seed, float* __restrict__ sink, int iters) {
    // 4 independent accumulators per wave, operands from registers
    unsigned s = seed[threadIdx.x & 63];
    bf16x8   a, b;
    for (int i = 0; i < 8; ++i) {
        s    = s * 1664525u + 1013904223u;
        a[i] = (short)((s >> 16) & (seed[64] ? 0x3fff : 0));  // small positive bf16 / f16 values, or all zeros
        s    = s * 1664525u + 1013904223u;
        b[i] = (short)((s >> 16) & (seed[64] ? 0x3fff : 0));
    }
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
        for (int r = 0; r < 16; ++r)
            acc[k][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (F16)
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[k], 0, 0, 0);
            else
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int k = 0; k < 4; ++k)
        for (int r = 0; r < 16; ++r)
            t += acc[k][r];
    if (t == 123.456f)
        sink[threadIdx.x] = t;
}

template<bool PACKED>
__global__ __launch_bounds__(256) void fma_rate(float* __restrict__ sink, int iters, float x0) {
    f32x2 v[8];
    for (int k = 0; k < 8; ++k)
        v[k] = f32x2{x0 + k, x0 - k};
    const f32x2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (PACKED)
                v[k] = __builtin_elementwise_fma(v[k], m, c);  // v_pk_fma_f32
            else {
                v[k].x = __builtin_fmaf(v[k].x, m.x, c.x);
                v[k].y = __builtin_fmaf(v[k].y, m.y, c.y);
            }
        }
    }
    float t = 0.f;
    for (int k = 0; k < 8; ++k)
        t += v[k].x + v[k].y;
    if (t == 123.456f)
        sink[threadIdx.x] = t;
}

static float time_ms(hipEvent_t a, hipEvent_t b) {
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t bytes = (size_t)2 << 30;
    f32x4 *      a, *b, *c;
    CK(hipMalloc((void**)&a, bytes));
    CK(hipMalloc((void**)&b, bytes));
    CK(hipMalloc((void**)&c, bytes));
    CK(hipMemset(b, 0, bytes));
    CK(hipMemset(c, 0, bytes));
    double copy = 0, tri = 0, cpk = 0, wr = 0, rd = 0, wp = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        CK(hipMemcpyAsync(a, b, bytes, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        if (rep)
            copy = std::max(copy, 2.0 * bytes / (time_ms(e0, e1) * 1e-3) / 1e9);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(triad, dim3(bytes / 16 / 1024), dim3(256), 0, 0, a, b, c, bytes / 16, 0.5f);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        if (rep)
            tri = std::max(tri, 3.0 * bytes / (time_ms(e0, e1) * 1e-3) / 1e9);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(copy4, dim3(bytes / 16 / 1024), dim3(256), 0, 0, a, b);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        if (rep)
            cpk = std::max(cpk, 2.0 * bytes / (time_ms(e0, e1) * 1e-3) / 1e9);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(fill4, dim3(bytes / 16 / 1024), dim3(256), 0, 0, a, 1.5f);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        if (rep)
            wr = std::max(wr, 1.0 * bytes / (time_ms(e0, e1) * 1e-3) / 1e9);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(read4, dim3(bytes / 16 / 1024), dim3(256), 0, 0, b, (float*)c);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        if (rep)
            rd = std::max(rd, 1.0 * bytes / (time_ms(e0, e1) * 1e-3) / 1e9);
        {
            const int n_rows = 51200, row_f4 = 2500, ppr = 625;  // 51200 x 10000 floats = 2.05 GB
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(fill_pieces, dim3(ppr * (n_rows / 256)), dim3(256), 0, 0, a, 2.5f, n_rows, row_f4, ppr);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            if (rep)
                wp = std::max(wp, (double)n_rows * row_f4 * 16 / (time_ms(e0, e1) * 1e-3) / 1e9);
        }
    }
    unsigned* seed;
    float*    sink;
    CK(hipMalloc((void**)&seed, 65 * 4));
    CK(hipMalloc((void**)&sink, 1024));
    std::vector<unsigned> hs(65);
    for (int i = 0; i < 64; ++i)
        hs[i] = 12345u + 977u * i;
    double mf[2][2] = {{0, 0}, {0, 0}};  // [f16][random]
    const int iters = 20000, blocks = p.multiProcessorCount * 2;  // 8 waves per CU = 2 per SIMD
    for (int f16 = 0; f16 < 2; ++f16)
        for (int rnd = 0; rnd < 2; ++rnd) {
            hs[64] = rnd;
            CK(hipMemcpy(seed, hs.data(), 65 * 4, hipMemcpyHostToDevice));
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (f16)
                    hipLaunchKernelGGL(mfma_rate<true>, dim3(blocks), dim3(256), 0, 0, seed, sink, iters);
                else
                    hipLaunchKernelGGL(mfma_rate<false>, dim3(blocks), dim3(256), 0, 0, seed, sink, iters);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                const double fl = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
                if (rep)
                    mf[f16][rnd] = std::max(mf[f16][rnd], fl / (time_ms(e0, e1) * 1e-3) / 1e12);
            }
        }
    double fma[2] = {0, 0};
    for (int pk = 0; pk < 2; ++pk)
        for (int rep = 0; rep < 3; ++rep) {
            const int it2 = 100000, bl = p.multiProcessorCount * 8;
            CK(hipEventRecord(e0, 0));
            if (pk)
                hipLaunchKernelGGL(fma_rate<true>, dim3(bl), dim3(256), 0, 0, sink, it2, 1.5f);
            else
                hipLaunchKernelGGL(fma_rate<false>, dim3(bl), dim3(256), 0, 0, sink, it2, 1.5f);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            const double fl = (double)bl * 256 * it2 * 16 * 2.0;
            if (rep)
                fma[pk] = std::max(fma[pk], fl / (time_ms(e0, e1) * 1e-3) / 1e12);
        }
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d, \"memory_clock_mhz\": %d, \"hbm_bytes\": %zu,\n"
           " \"hbm_memcpy_GBps\": %.0f, \"hbm_copy_kernel_GBps\": %.0f, \"hbm_triad_GBps\": %.0f, \"hbm_write_only_GBps\": %.0f, \"hbm_read_only_GBps\": %.0f, \"hbm_write_64B_pieces_GBps\": %.0f, \"datasheet_hbm_GBps\": 8000,\n"
           " \"mfma_bf16_zero_TFLOPs\": %.0f, \"mfma_bf16_random_TFLOPs\": %.0f, \"mfma_f16_zero_TFLOPs\": %.0f, \"mfma_f16_random_TFLOPs\": %.0f, "
           "\"datasheet_bf16_dense_TFLOPs\": 2500,\n"
           " \"fma_f32_TFLOPs\": %.1f, \"pk_fma_f32_TFLOPs\": %.1f, \"datasheet_f32_vector_TFLOPs\": 157.3}\n",
           p.name, p.multiProcessorCount, p.clockRate / 1000, p.memoryClockRate / 1000, (size_t)p.totalGlobalMem, copy, cpk, tri, wr, rd, wp, mf[0][0], mf[0][1],
           mf[1][0], mf[1][1], fma[0], fma[1]);
    return 0;
}
