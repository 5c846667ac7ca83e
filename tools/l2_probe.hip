// l2_probe.hip -- what does a CU pull per second out of (a) its XCD's L2, (b) a 12.6 MB matrix every XCD streams, (c) 100 MB, by path?
// (round-5 review, "settle the 47 GB/s question": the guide prices L2 at ~34.5 TB/s = 135 GB/s per CU; tools/gather_probe.hip measured
// 12 TB/s on a working set of 8 MB per XCD, i.e. twice an XCD's L2.)
// One 512-thread workgroup per CU, all 256 CUs; every wave-instruction moves 1 KB (64 lanes x 16 B, contiguous, coalesced):
//   path V   global_load_dwordx4 -> VGPR (8 per wave in flight, xor-ed into a sink)
//   path D   global_load_lds_dwordx4 -> LDS (two groups of 8 per wave, counted vmcnt: 8..16 KB per wave in flight, 128 KB of LDS)
// source  a: 2 MB per XCD (block b reads region b % 8 -- the dispatcher's round robin; XCC_ID mismatches are counted), re-read 64 times
//         b: ONE 12.6 MB region read by every workgroup (the f16mx image of a 2048 x 2048 layer), CU c starts at chunk c
//         c: ONE 100 MB region, the same walk
// Hit rates come from a second run under `rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum` (tools/pmc_summary.py).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/build/l2_probe tools/l2_probe.hip ; tools/build/l2_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int NW = 8, G = 8, CHUNK = NW * G * 1024;  // a workgroup moves 64 KB per group

template<int PATH>
__global__ __launch_bounds__(512) void pull(const char* __restrict__ src, size_t region_bytes, int per_xcd, int n_groups, int* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int      lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int      xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
    const char*    base = src + (per_xcd ? (size_t)xcd * region_bytes : 0);
    const int      n_chunks = (int)(region_bytes / CHUNK);
    const unsigned voff = lane * 16u;
    v4i            acc = {0, 0, 0, 0};
    if (threadIdx.x == 0 && (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7) != xcd)   // HW_REG_XCC_ID
        atomicAdd(sink + 1, 1);
    int c = (per_xcd ? cu : blockIdx.x) % n_chunks;
    if (PATH == 0) {
        for (int g = 0; g < n_groups; ++g) {
            const char* p = base + (size_t)c * CHUNK + wave * (G * 1024) + voff;
            v4i         a[G];
#pragma unroll
            for (int u = 0; u < G; ++u)
                a[u] = *(const v4i*)(p + u * 1024);
#pragma unroll
            for (int u = 0; u < G; ++u)
                acc ^= a[u];
            c = c + 1 == n_chunks ? 0 : c + 1;
        }
    }
    else {
        const unsigned lds_base = (unsigned)(uintptr_t)lds;
        auto           issue    = [&](int g) {
            const char* p = base + (size_t)c * CHUNK + wave * (G * 1024);
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + ((g & 1) * NW * G + wave * G + u) * 1024));
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(p + u * 1024), "s"(dst) : "memory");
            }
            c = c + 1 == n_chunks ? 0 : c + 1;
        };
        issue(0);
        for (int g = 1; g < n_groups; ++g) {
            issue(g);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");   // group g - 1 has landed
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc.x = *(const int*)(lds + threadIdx.x * 4);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678)
        sink[0] = 1;
}

template<int PATH>
static void run(const char* what, const char* src, size_t region, int per_xcd, int n_cu) {
    auto k = pull<PATH>;
    const int lds_bytes = PATH ? 2 * NW * G * 1024 : 0;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    static int* sink = nullptr;
    if (!sink) {
        hipMalloc(&sink, 64);
        hipMemset(sink, 0, 64);
    }
    const int n_groups = 2048;   // 128 MB per CU and launch
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(n_cu), dim3(512), lds_bytes, 0, src, region, per_xcd, 256, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(n_cu), dim3(512), lds_bytes, 0, src, region, per_xcd, n_groups, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    int h[2];
    hipMemcpy(h, sink, 8, hipMemcpyDeviceToHost);
    const double bytes = (double)n_groups * CHUNK;
    printf("  %-34s %-30s %8.3f ms  %6.1f GB/s per CU  %6.2f TB/s chip   (XCC_ID != block %% 8: %d)\n", what, PATH ? "D global_load_lds_dwordx4 -> LDS" : "V global_load_dwordx4 -> VGPR", ms,
           bytes / (ms * 1e-3) / 1e9, bytes * n_cu / (ms * 1e-3) / 1e12, h[1]);
    hipMemset(sink, 0, 64);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    char*     src  = nullptr;
    const size_t cap = (size_t)100 << 20;
    hipMalloc(&src, cap);
    hipMemset(src, 1, cap);
    printf("l2_probe: %d CUs, one 512-thread workgroup each, 1 KB per wave-instruction, 128 MB per CU and launch\n", n_cu);
    const size_t w = (size_t)192 * CHUNK;   // 12.58 MB: 2048 x 2048 x 3 B
    run<0>("a 2 MB per XCD, re-read", src, (size_t)2 << 20, 1, n_cu);
    run<1>("a 2 MB per XCD, re-read", src, (size_t)2 << 20, 1, n_cu);
    run<0>("a' 1 MB per XCD, re-read", src, (size_t)1 << 20, 1, n_cu);
    run<1>("a' 1 MB per XCD, re-read", src, (size_t)1 << 20, 1, n_cu);
    run<0>("b 12.6 MB shared by all XCDs", src, w, 0, n_cu);
    run<1>("b 12.6 MB shared by all XCDs", src, w, 0, n_cu);
    run<0>("c 100 MB shared by all XCDs", src, cap / CHUNK * CHUNK, 0, n_cu);
    run<1>("c 100 MB shared by all XCDs", src, cap / CHUNK * CHUNK, 0, n_cu);
    return 0;
}
