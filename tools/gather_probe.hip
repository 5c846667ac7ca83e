// gather_probe.hip -- how fast can a CU pull scattered 256-byte rows out of L2, by request shape?  (tied_pruned_kernel's access)
// Every wave reads `rows` pseudo-random rows of a table slice that fits L2; variants differ in how a row is fetched:
//   0: 64 lanes x dword (one row per instruction)      1: 32 lanes x dwordx2 (lanes 32..63 idle)
//   2: 16 lanes x dwordx4 (lanes 16..63 idle)          3: 64 lanes x dwordx4 = 4 rows per instruction
//   4: 64 lanes x dwordx2 = 2 rows per instruction
// hipcc --offload-arch=gfx950 -O3 -o tools/build/gather_probe tools/gather_probe.hip ; tools/build/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

template <int VAR, int PF>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ tab, const uint32_t* __restrict__ idx, int rows, int row_floats, int tile_stride,
                                             float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w    = blockIdx.x * 4 + wave;
    const int tile = (blockIdx.x >> 6) & 7;  // a few tiles at a time, like the kernel
    const uint32_t* my = idx + (size_t)w * rows;
    const float*    base = tab + (size_t)tile * tile_stride;
    float acc = 0.f;
    for (int i = 0; i < rows; i += PF * (VAR == 3 ? 4 : VAR == 4 ? 2 : 1)) {
        if (VAR == 0) {
            float a[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u)
                a[u] = base[(size_t)my[i + u] * row_floats + lane];
#pragma unroll
            for (int u = 0; u < PF; ++u)
                acc += a[u];
        }
        else if (VAR == 1) {
            float2 a[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u)
                a[u] = lane < 32 ? *(const float2*)(base + (size_t)my[i + u] * row_floats + lane * 2) : make_float2(0, 0);
#pragma unroll
            for (int u = 0; u < PF; ++u)
                acc += a[u].x + a[u].y;
        }
        else if (VAR == 2) {
            float4 a[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u)
                a[u] = lane < 16 ? *(const float4*)(base + (size_t)my[i + u] * row_floats + lane * 4) : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < PF; ++u)
                acc += a[u].x + a[u].y + a[u].z + a[u].w;
        }
        else if (VAR == 3) {
            float4 a[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u)
                a[u] = *(const float4*)(base + (size_t)my[i + 4 * u + (lane >> 4)] * row_floats + (lane & 15) * 4);
#pragma unroll
            for (int u = 0; u < PF; ++u)
                acc += a[u].x + a[u].y + a[u].z + a[u].w;
        }
        else {
            float2 a[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u)
                a[u] = *(const float2*)(base + (size_t)my[i + 2 * u + (lane >> 5)] * row_floats + (lane & 31) * 2);
#pragma unroll
            for (int u = 0; u < PF; ++u)
                acc += a[u].x + a[u].y;
        }
    }
    out[(size_t)w * 64 + lane] = acc;
}

// scattered: every active lane reads 4 bytes of its own random row; `active` lanes per instruction take part
template <int PF>
__global__ __launch_bounds__(256) void scatter_probe(const float* __restrict__ tab, const uint32_t* __restrict__ idx, int iters, int row_floats,
                                                     int active, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w    = blockIdx.x * 4 + wave;
    const uint32_t* my = idx + (size_t)w * 64;
    uint32_t k = my[lane];
    float acc = 0.f;
    for (int i = 0; i < iters; i += PF) {
        float a[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            a[u] = 0.f;
            if (lane < active)
                a[u] = tab[(size_t)((k + 97u * (i + u)) & 4095u) * row_floats + lane];
        }
#pragma unroll
        for (int u = 0; u < PF; ++u)
            acc += a[u];
    }
    out[(size_t)w * 64 + lane] = acc;
}

int main(int argc, char** argv) {
    const int K = 4096, row_floats = 10048, n_waves = 40192, rows = 64, reps = 20;
    float* tab; uint32_t* idx; float* out;
    hipMalloc(&tab, (size_t)K * row_floats * 4);
    hipMemset(tab, 0, (size_t)K * row_floats * 4);
    std::vector<uint32_t> h((size_t)n_waves * rows);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) % K; }
    hipMalloc(&idx, h.size() * 4);
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)n_waves * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(n_waves / 4), dim3(256), 0, 0, tab, idx, rows, row_floats, 64, out);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(n_waves / 4), dim3(256), 0, 0, tab, idx, rows, row_floats, 64, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        const double bytes = (double)n_waves * rows * 256;
        printf("%-44s %8.1f us  %7.1f GB/s  (%5.1f GB/s per CU)\n", name, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 256);
    };
    run("0: dword x 64 lanes, 16 in flight", probe<0, 16>);
    run("0: dword x 64 lanes, 32 in flight", probe<0, 32>);
    run("0: dword x 64 lanes, 8 in flight", probe<0, 8>);
    run("1: dwordx2 x 32 lanes, 16 in flight", probe<1, 16>);
    run("2: dwordx4 x 16 lanes, 16 in flight", probe<2, 16>);
    run("3: dwordx4 x 64 lanes (4 rows), 4 in flight", probe<3, 4>);
    run("3: dwordx4 x 64 lanes (4 rows), 8 in flight", probe<3, 8>);
    run("3: dwordx4 x 64 lanes (4 rows), 16 in flight", probe<3, 16>);
    run("4: dwordx2 x 64 lanes (2 rows), 8 in flight", probe<4, 8>);
    run("4: dwordx2 x 64 lanes (2 rows), 16 in flight", probe<4, 16>);
    auto run_s = [&](int active) {
        const int iters = 24;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(scatter_probe<8>, dim3(n_waves / 4), dim3(256), 0, 0, tab, idx, iters, row_floats, active, out);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(scatter_probe<8>, dim3(n_waves / 4), dim3(256), 0, 0, tab, idx, iters, row_floats, active, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        const double instr_per_cu = (double)n_waves * iters / 256;
        printf("scattered dword, %2d active lanes, 24 instr/wave: %8.1f us  = %6.1f cycles (2.4 GHz) per wave instruction per CU\n", active, ms * 1e3,
               ms * 1e-3 * 2.4e9 / instr_per_cu);
    };
    run_s(64); run_s(32); run_s(16); run_s(4); run_s(1);
    return 0;
}
