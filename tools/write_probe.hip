// tools/write_probe.hip -- HBM write rate of a [rows x 10000] f32 matrix written in row pieces of 64..1024 bytes (the store
// pattern of a scorer that produces 16..256 mixtures of a frame at a time), with the pieces of one row dealt to workgroups
// round robin (neighbouring pieces on different XCDs) or in contiguous ranges per XCD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/write_probe.hip -o tools/build/write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// block = 256 rows x one piece of P float4; mode 0: piece = bx % ppr; mode 1: XCD-contiguous piece ranges; nt: nontemporal
template<int P>
__global__ __launch_bounds__(256) void fill(f32x4* __restrict__ a, int n_rows, int row_f4, int ppr, int mode, int nt) {
    int bx = blockIdx.x, piece, rb;
    if (mode == 0) {
        piece = bx % ppr;
        rb    = bx / ppr;
    }
    else {
        const int per = (ppr + 7) / 8, x = bx % (per * 8);
        rb    = bx / (per * 8);
        piece = (x & 7) * per + (x >> 3);
        if (piece >= ppr)
            return;
    }
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (int e = threadIdx.x; e < 256 * P; e += 256) {
        const int row = rb * 256 + e / P, c = e % P;
        f32x4*    dst = a + (size_t)row * row_f4 + piece * P + c;
        if (row < n_rows && piece * P + c < row_f4) {
            if (nt)
                __builtin_nontemporal_store(v, dst);
            else
                *dst = v;
        }
    }
}

template<int P>
void run(f32x4* a, int n_rows, int row_f4) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int ppr = (row_f4 + P - 1) / P;
    for (int mode = 0; mode < 2; ++mode)
        for (int nt = 0; nt < 2; ++nt) {
            const int gx = mode ? (ppr + 7) / 8 * 8 : ppr;
            float     best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(fill<P>, dim3(gx * (n_rows / 256)), dim3(256), 0, 0, a, n_rows, row_f4, ppr, mode, nt);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best)
                    best = ms;
            }
            printf("piece %4d B  %-14s %-3s  %6.0f GB/s\n", P * 16, mode ? "xcd-contiguous" : "round-robin", nt ? "nt" : "", (double)n_rows * row_f4 * 16 / (best * 1e-3) / 1e9);
        }
}

int main() {
    const int n_rows = 65536, row_f4 = 2500;
    f32x4*    a;
    CK(hipMalloc((void**)&a, (size_t)n_rows * row_f4 * 16));
    run<4>(a, n_rows, row_f4);
    run<8>(a, n_rows, row_f4);
    run<16>(a, n_rows, row_f4);
    run<32>(a, n_rows, row_f4);
    run<64>(a, n_rows, row_f4);
    return 0;
}
