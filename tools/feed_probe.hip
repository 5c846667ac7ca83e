// feed_probe.hip -- how fast can a CU pull the operand stream of gemm_mx_kernel (48 KB per K-tile and workgroup) out of L2, by path?
// One persistent 512-thread workgroup per CU walks K-tiles; every K-tile each wave moves 6 x 1 KB (64 lanes x 16 B, contiguous):
//   mode 0  global_load_lds_dwordx4 straight into a 3-stage LDS ring, counted vmcnt + one barrier per K-tile (the kernel's path)
//   mode 1  global_load_dwordx4 into registers, ds_write_b128 into the ring one K-tile later (two K-tiles of registers in flight)
//   mode 2  global_load_dwordx4 into registers only (what L2 -> CU delivers when nothing is written to LDS)
//   mode 3  half of the pieces as in mode 0, half as in mode 1
// `mfma` adds 24 v_mfma_f32_32x32x16_f16 per wave and K-tile on register operands (the matrix work of the real kernel, clocks included);
// `span` = K-tiles of source per workgroup before it wraps (8: 384 KB per workgroup, L2 / MALL resident; shared = all workgroups of a
// launch read the same 1.5 MB).  Prints GB/s per CU and TB/s per chip.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/build/feed_probe tools/feed_probe.hip ; tools/build/feed_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef float    v4f __attribute__((ext_vector_type(4)));
typedef int      v4i __attribute__((ext_vector_type(4)));
typedef int      v8i __attribute__((ext_vector_type(8)));  // a register tuple the asm constraints accept (HIP's float4 is a struct)

constexpr int PPW = 6, KT_BYTES = 8 * PPW * 1024, STAGES = 3;

template<int MODE, bool MFMA, int MISS = 0, int PF = 0, int PIPE = 0, int PSPAN = 8, bool MXMIX = false>
__global__ __launch_bounds__(512) void feed(const char* __restrict__ src, int n_kt, int span, int shared, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int      lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char*    base = src + (shared ? 0 : (size_t)blockIdx.x * span * KT_BYTES);
    const unsigned voff = lane * 16u, lds_base = (unsigned)(uintptr_t)lds;
    f32x16         acc[8];
    f16x8          fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r)
            acc[i][r] = 0.f;
    v4f r0[PPW], r1[PPW];
    f16x8 fr0[18], fr1[18];
    for (int r = 0; r < 18; ++r)
        fr0[r] = fr1[r] = fa;
    float  s = 0.f;
    // MISS > 0 (with shared = 1): the last MISS of a wave's 6 pieces come from a region of its own (384 KB per workgroup: L2 misses
    // served by the Infinity Cache) -- the real kernel's mix is 77 % L2 hits; PF > 0: those lines are touched PF K-tiles ahead with
    // one dword per 128-byte line (a software prefetch into L2; wave 0 only, one load per 8 KB)
    const char* priv = src + (size_t)(32 + blockIdx.x * PSPAN) * KT_BYTES;  // PSPAN 8: 384 KB per workgroup (Infinity Cache); 60: 2.9 MB (HBM)
    auto        dma  = [&](int kt, int q) {
        const char* p = base + (size_t)((kt + (shared ? blockIdx.x : 0)) % span) * KT_BYTES + (q * 8 + wave) * 1024;
        if (MISS > 0 && q >= PPW - MISS)
            p = priv + (size_t)(kt % PSPAN) * KT_BYTES + (q * 8 + wave) * 1024;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (kt % STAGES) * KT_BYTES + (q * 8 + wave) * 1024));
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(p), "s"(dst) : "memory");
    };
    float pf_sink = 0.f;
    auto  prefetch = [&](int kt) {  // the MISS * 8 KB of K-tile kt that will miss: 64 lines per load
        if (PF > 0 && MISS > 0 && wave == 0) {
#pragma unroll
            for (int j = 0; j < MISS; ++j) {
                const char* p = priv + (size_t)(kt % PSPAN) * KT_BYTES + ((PPW - MISS + j) * 8) * 1024;
                float       v;
                asm volatile("global_load_dword %0, %1, %2" : "=&v"(v) : "v"(lane * 128u), "s"(p) : "memory");
                pf_sink += v;   // never waited for explicitly: it is older than the pieces the counted wait covers
            }
        }
    };
    // register loads through inline assembly too: the compiler does not count the LDS-DMA pieces of an asm statement, its own
    // s_waitcnt vmcnt(N) for a load it knows would be wrong beside them; all waits are explicit (wait<N> ties them to the registers)
    auto ld = [&](int kt, int q) {
        const char* p = base + (size_t)((kt + (shared ? blockIdx.x : 0)) % span) * KT_BYTES + (q * 8 + wave) * 1024;
        v4f         v;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v) : "v"(voff), "s"(p) : "memory");
        return v;
    };
    constexpr int NDMA = MODE == 0 ? PPW : MODE == 3 ? PPW / 2 : 0;  // pieces per wave through LDS-DMA
    constexpr int NREG = PPW - NDMA;
    // prologue: two K-tiles in flight
#pragma unroll
    for (int q = 0; q < NDMA; ++q)
        dma(0, q);
#pragma unroll
    for (int q = 0; q < NREG; ++q)
        r0[q] = ld(0, NDMA + q);
#pragma unroll
    for (int q = 0; q < NDMA; ++q)
        dma(1, q);
#pragma unroll
    for (int q = 0; q < NREG; ++q)
        r1[q] = ld(1, NDMA + q);
    auto body = [&](v4f (&regs)[PPW], int kt) {
        // K-tile kt has landed: every mode keeps 2 x PPW operations in flight; the older K-tile's PPW are done at vmcnt(PPW)
        if constexpr (NREG == 0) {
            if (PF > 0 && MISS > 0 && wave == 0)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 2 * MISS) : "memory");  // + the prefetches issued behind K-tiles kt and kt + 1
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        }
        else if constexpr (NREG == 3)
            asm volatile("s_waitcnt vmcnt(%3)" : "+v"(regs[0]), "+v"(regs[1]), "+v"(regs[2]) : "n"(PPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%6)" : "+v"(regs[0]), "+v"(regs[1]), "+v"(regs[2]), "+v"(regs[3]), "+v"(regs[4]), "+v"(regs[5]) : "n"(PPW) : "memory");
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int q = 0; q < NREG; ++q)
                *(v4f*)(lds + (kt % STAGES) * KT_BYTES + ((NDMA + q) * 8 + wave) * 1024 + voff) = regs[q];
        }
        else if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < NREG; ++q)
                s += regs[q].x + regs[q].w;
        }
        __builtin_amdgcn_s_barrier();
        // refill
#pragma unroll
        for (int q = 0; q < NDMA; ++q)
            dma(kt + 2, q);
        prefetch(kt + 2 + PF);
#pragma unroll
        for (int q = 0; q < NREG; ++q)
            regs[q] = ld(kt + 2, NDMA + q);
        if (PIPE == 0) {
            if (MODE != 2) {  // consume: 18 fragment-sized reads per wave like the kernel
                const char* st = lds + (kt % STAGES) * KT_BYTES;
#pragma unroll
                for (int r = 0; r < 18; ++r) {
                    const f16x8 v = *(const f16x8*)(st + ((r * 8 + wave) % 48) * 1024 + voff);
                    fa[r & 7] += v[0];
                }
            }
            if (MFMA) {
#pragma unroll
                for (int m = 0; m < 24; ++m)
                    acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[m & 7], 0, 0, 0);
            }
        }
        else {
            // PIPE 1: 18 fragment registers, 24 matrix instructions on them right behind the reads (the kernel's order)
            // PIPE 2: the reads of this period fill the OTHER register set; the matrix instructions run on the set read one period
            //         earlier -- they can start at the barrier, the LDS round trip hides behind them (needs 72 more registers)
            const char* st = lds + (kt % STAGES) * KT_BYTES;
            f16x8 (&dst)[18] = (PIPE == 2 && (kt & 1)) ? fr1 : fr0;
            f16x8 (&src)[18] = (PIPE == 2) ? ((kt & 1) ? fr0 : fr1) : fr0;
#pragma unroll
            for (int r = 0; r < 18; ++r)
                dst[r] = *(const f16x8*)(st + ((r * 8 + wave) % 48) * 1024 + voff);
            if (MFMA) {
                if (MXMIX) {  // the kernel's mix: 16 f16 products (32 cycles each) + 8 block-scaled fp6 x fp6 products (64 cycles), operands built from the fragments
#pragma unroll
                    for (int m = 0; m < 16; ++m)
                        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(src[m % 12], src[12 + m % 6], acc[m & 3], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const v8i av = __builtin_shufflevector(__builtin_bit_cast(v4i, src[m % 12]), __builtin_bit_cast(v4i, src[(m + 1) % 12]), 0, 1, 2, 3, 4, 5, 6, 7);
                        const v8i bv = __builtin_shufflevector(__builtin_bit_cast(v4i, src[12 + m % 6]), __builtin_bit_cast(v4i, src[12 + (m + 1) % 6]), 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[m & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[m & 3], 2, 2, 0, 127, 0, 127);
                    }
                }
                else {
#pragma unroll
                    for (int m = 0; m < 24; ++m)
                        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(src[m % 12], src[12 + m % 6], acc[m & 3], 0, 0, 0);
                }
            }
        }
    };
    for (int kt = 0; kt < n_kt; kt += 2) {
        body(r0, kt);
        body(r1, kt + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 0; i < 8; ++i)
        s += acc[i][0] + acc[i][7];
    s += (float)fa[0] + pf_sink + (float)fr0[3][1] + (float)fr1[5][2];
    if (s == 123.456f)
        sink[threadIdx.x] = s;
}

template<int MODE, bool MFMA, int MISS = 0, int PF = 0, int PIPE = 0, int PSPAN = 8, bool MXMIX = false>
static void run(const char* name, const char* src, int span, int shared, float* sink, int n_cu) {
    const int n_kt = 4096, lds_bytes = STAGES * KT_BYTES;
    auto      k    = feed<MODE, MFMA, MISS, PF, PIPE, PSPAN, MXMIX>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(n_cu), dim3(512), lds_bytes, 0, src, 256, span, shared, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(n_cu), dim3(512), lds_bytes, 0, src, n_kt, span, shared, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)n_kt * KT_BYTES;
    printf("  %-58s span %3d %s  %7.3f ms  %6.1f GB/s per CU  %6.2f TB/s chip  %7.1f ns per K-tile%s\n", name, span, shared ? "shared " : "private", ms,
           bytes / (ms * 1e-3) / 1e9, bytes * n_cu / (ms * 1e-3) / 1e12, ms * 1e6 / n_kt, hipGetLastError() == hipSuccess ? "" : "  (launch error)");
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int    n_cu = prop.multiProcessorCount;
    const size_t cap  = (size_t)n_cu * 64 * KT_BYTES + 40 * KT_BYTES;
    char*        src  = nullptr;
    float*       sink = nullptr;
    hipMalloc(&src, cap);
    hipMemset(src, 1, cap);
    hipMalloc(&sink, 4096);
    printf("feed_probe: %d CUs, 48 KB per K-tile and workgroup, 4096 K-tiles per workgroup\n", n_cu);
    for (int shared = 1; shared >= 0; --shared)
        for (int span : {8, 64}) {
            if (shared && span == 64)
                continue;
            const int sp = shared ? 32 : span;
            printf("source: %s\n", shared ? "one 1.5 MB region read by every workgroup (L2 hits)" : span == 8 ? "384 KB per workgroup (12 MB per XCD: Infinity Cache)"
                                                                                                          : "3 MB per workgroup (HBM / Infinity Cache)");
            run<0, false>("0 LDS-DMA", src, sp, shared, sink, n_cu);
            run<1, false>("1 global_load -> registers -> ds_write_b128", src, sp, shared, sink, n_cu);
            run<3, false>("3 half LDS-DMA, half through registers", src, sp, shared, sink, n_cu);
            run<2, false>("2 global_load -> registers only", src, sp, shared, sink, n_cu);
            run<0, true>("0 LDS-DMA + 24 MFMA per wave and K-tile", src, sp, shared, sink, n_cu);
            run<1, true>("1 registers -> ds_write + 24 MFMA", src, sp, shared, sink, n_cu);
            run<3, true>("3 half / half + 24 MFMA", src, sp, shared, sink, n_cu);
            run<2, true>("2 registers only + 24 MFMA", src, sp, shared, sink, n_cu);
            if (shared) {
                run<0, true, 0, 0, 1>("0 LDS-DMA + MFMA on 18 fragment registers, reads then products", src, sp, shared, sink, n_cu);
                run<0, true, 0, 0, 2>("0 LDS-DMA + MFMA, products on the set read one period earlier", src, sp, shared, sink, n_cu);
                run<0, true, 1, 0, 1, 60>("0 LDS-DMA + MFMA (fragment form), 1 of 6 pieces from HBM", src, sp, shared, sink, n_cu);
                run<0, true, 2, 0, 1, 60>("0 LDS-DMA + MFMA (fragment form), 2 of 6 pieces from HBM", src, sp, shared, sink, n_cu);
                run<0, true, 2, 0, 2, 60>("0 LDS-DMA + MFMA (pipelined form), 2 of 6 pieces from HBM", src, sp, shared, sink, n_cu);
                run<0, true, 2, 0, 1, 8>("0 LDS-DMA + MFMA (fragment form), 2 of 6 from the Infinity Cache", src, sp, shared, sink, n_cu);
                run<0, true, 0, 0, 1, 8, true>("0 LDS-DMA + 16 f16 + 8 scaled fp6 products (fragment form), L2 hits", src, sp, shared, sink, n_cu);
                run<0, true, 2, 0, 1, 60, true>("0 LDS-DMA + 16 f16 + 8 scaled fp6 products, 2 of 6 pieces from HBM", src, sp, shared, sink, n_cu);
                run<0, true, 2, 0, 2, 60, true>("0 ... the same, pipelined form", src, sp, shared, sink, n_cu);
                run<2, true, 0, 0, 1, 8, true>("2 registers only + 16 f16 + 8 scaled fp6 products (matrix-bound)", src, sp, shared, sink, n_cu);
                run<0, true, 1>("0 LDS-DMA + MFMA, 1 of 6 pieces misses L2", src, sp, shared, sink, n_cu);
                run<0, true, 2>("0 LDS-DMA + MFMA, 2 of 6 pieces miss L2", src, sp, shared, sink, n_cu);
                run<0, true, 2, 3>("0 LDS-DMA + MFMA, 2 of 6 miss, prefetched 3 K-tiles ahead", src, sp, shared, sink, n_cu);
                run<0, true, 2, 6>("0 LDS-DMA + MFMA, 2 of 6 miss, prefetched 6 K-tiles ahead", src, sp, shared, sink, n_cu);
                run<0, false, 2>("0 LDS-DMA, 2 of 6 pieces miss L2", src, sp, shared, sink, n_cu);
                run<0, false, 2, 3>("0 LDS-DMA, 2 of 6 miss, prefetched 3 K-tiles ahead", src, sp, shared, sink, n_cu);
            }
        }
    return 0;
}
