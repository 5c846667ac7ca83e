// tools/gemm_probe.hip -- experiment driver for the bf16 GEMM kernel (not part of the library, not a test).
//   make -C rasr_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I rasr_amd/csrc \
//        tools/gemm_probe.hip rasr_amd/csrc/build/api.o rasr_amd/csrc/build/stats.o -o gpurun_out/gemm_probe
// Runs the output layer of the bench network (2048 -> 10000, 32768 frames) under the schedule variants named on the
// command line (VAR bit masks, see gemm_bf16_kernel) and prints the average launch time; `trace` adds the per-tile
// time stamps of variant 512 (skew between the workgroups of an XCD).
#include "../rasr_amd/csrc/ffnn.hip"

#include <cstdio>
#include <random>
#include <string>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

struct Problem {
    int             N = 10000, K = 2048, T = 32768, Npad, Tpad;
    amx::bf16_t *   W, *X;      // rows [hi plane | lo plane], 2 K columns (plain bf16 launches use the hi plane only)
    float *         bias, *out, *pmin;
    unsigned*       pidx;
    unsigned long long* trace;
    unsigned*       sync;
    int             n_cu;
    int             gt = 0, gn = 0;
    int             pad = 0;    // PROBE_PAD: extra elements per operand row (row stride 2 K + pad)
    amx::GemmLd ld(bool x3) const { return amx::GemmLd{2 * K + pad, x3 ? K : 0, 2 * K + pad, x3 ? K : 0, N, 0}; }
};

template<class C, int VAR>
float run(Problem& p, int iters, bool quiet = false) {
    auto          k   = amx::gemm_bf16_kernel<C, AMX_ACT_NONE, true, VAR>;
    const int     ntn = p.Npad / C::BN, ntt = p.Tpad / C::BT;
    constexpr int lds = amx::gemm_scratch_bytes<C, true>() + C::BN * 4;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int per_cu = std::max(1, (160 * 1024) / lds);
    int       grid   = std::min(ntn * ntt, per_cu * p.n_cu) & ~7;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float total = 0;
    for (int it = -2; it < iters; ++it) {
        CK(hipMemsetAsync(p.sync, 0, 8 * 64 * 4, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(C::THREADS), lds, 0, p.W, p.X, p.bias, (void*)p.out, p.K, p.ld(C::X3), p.N, p.T, ntn, ntn * ntt, p.gt, p.gn,
                           p.pmin, p.pidx, p.Tpad);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 0)
            total += ms;
    }
    const float ms = total / iters;
    if (!quiet) printf("%s var %4d  group %dx%d  tile %dx%dx%d  grid %d  %.4f ms  %.0f TFLOP/s executed\n", C::X3 ? "x3  " : "bf16", VAR, p.gt, p.gn, C::BN, C::BT, C::BKC, grid, ms,
                       (C::X3 ? 3.0 : 1.0) * 2.0 * p.N * p.K * p.T / ms * 1e-9);
    fflush(stdout);
    return ms;
}

template<class C, int DBG>
float run_pipe(Problem& p, int iters, bool quiet = false) {
    using P       = amx::PipeLds<C, true>;
    auto      k   = amx::gemm_bf16_pipe_kernel<C, AMX_ACT_NONE, true, DBG>;
    const int ntn = p.Npad / C::BN, ntt = p.Tpad / C::BT;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, P::BYTES));
    int        grid = std::min(ntn * ntt, p.n_cu) & ~7;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float total = 0;
    for (int it = -2; it < iters; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(C::THREADS), P::BYTES, 0, p.W, p.X, p.bias, (void*)p.out, p.K, p.ld(C::X3), p.N, p.T, ntn, ntn * ntt,
                           p.gt, p.gn, 1, p.pmin, p.pidx, p.Tpad);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 0)
            total += ms;
    }
    const float ms = total / iters;
    if (!quiet) printf("%s pipe %3d  group %dx%d  tile %dx%dx%d  grid %d  %.4f ms  %.0f TFLOP/s executed\n", C::X3 ? "x3  " : "bf16", DBG, p.gt, p.gn, C::BN, C::BT, C::BKC, grid, ms,
                       (C::X3 ? 3.0 : 1.0) * 2.0 * p.N * p.K * p.T / ms * 1e-9);
    fflush(stdout);
    return ms;
}

void dump_trace(Problem& p, size_t trace_n) {
    std::vector<unsigned long long> tr(trace_n);
    CK(hipMemcpy(tr.data(), p.trace, trace_n * 8, hipMemcpyDeviceToHost));
    const int grid = 256, steps = (p.Npad / 256) * (p.Tpad / 256) / grid;
    unsigned long long t00 = ~0ull;
    for (int b = 0; b < grid; ++b) t00 = std::min(t00, tr[(size_t)b * 256]);
    int xcc_mismatch = 0;
    for (int b = 0; b < grid; ++b) xcc_mismatch += ((int)tr[(size_t)b * 256 + 3] & 15) != (b & 7);
    printf("workgroups whose XCC_ID != blockIdx %% 8: %d of %d\n", xcc_mismatch, grid);
    printf("step: per XCD 0 start skew (us) | all: start min..max, kloop avg, epilogue avg (us)   [100 MHz clock]\n");
    for (int st = 0; st < steps; ++st) {
        double smin = 1e30, smax = 0, kl = 0, ep = 0, x0min = 1e30, x0max = 0;
        for (int b = 0; b < grid; ++b) {
            const unsigned long long* r = &tr[((size_t)b * 64 + st) * 4];
            double s0 = (r[0] - t00) * 0.01, s1 = (r[1] - t00) * 0.01, s2 = (r[2] - t00) * 0.01;
            smin = std::min(smin, s0); smax = std::max(smax, s0);
            if ((b & 7) == 0) { x0min = std::min(x0min, s0); x0max = std::max(x0max, s0); }
            kl += s1 - s0; ep += s2 - s1;
        }
        if (st == steps - 1) {
            printf("end of last tile per XCD (us): ");
            for (int x = 0; x < 8; ++x) {
                double mn = 1e30, mx = 0;
                for (int b = x; b < grid; b += 8) {
                    double e = (tr[((size_t)b * 64 + st) * 4 + 2] - t00) * 0.01;
                    mn = std::min(mn, e); mx = std::max(mx, e);
                }
                printf(" x%d %.0f..%.0f", x, mn, mx);
            }
            printf("\n");
        }
        printf("%2d: xcd0 skew %6.1f | start %8.1f..%8.1f  kloop %6.1f  epi %6.1f\n", st, x0max - x0min, smin, smax, kl / grid, ep / grid);
    }
}

// arguments: <variant>[:<GT>x<GN>] ...   variants: see the table in main(); a leading 'x' selects the split-bf16 kernels
int main(int argc, char** argv) {
    Problem p;
    if (getenv("PROBE_N")) p.N = atoi(getenv("PROBE_N"));  // small problems: PROBE_N=2048 PROBE_T=1024 ... s0 s128 s136 s144 s192 t0 t128
    if (getenv("PROBE_T")) p.T = atoi(getenv("PROBE_T"));
    if (getenv("PROBE_K")) p.K = atoi(getenv("PROBE_K"));
    p.Npad = (p.N + 255) / 256 * 256;
    p.Tpad = p.T;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    p.n_cu = prop.multiProcessorCount;
    if (getenv("PROBE_PAD")) p.pad = atoi(getenv("PROBE_PAD"));
    const size_t rs = (size_t)2 * p.K + p.pad;
    std::vector<amx::bf16_t> w((size_t)p.Npad * rs), x((size_t)p.Tpad * rs);
    std::mt19937             rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    const bool relu = getenv("PROBE_RELU") != nullptr;  // hidden activations after ReLU: half the operand is zero
    auto fill = [&](std::vector<amx::bf16_t>& v, size_t rows, float scale, bool rl) {
        for (size_t r = 0; r < rows; ++r)
            for (int k = 0; k < p.K; ++k) {
                float f = nd(rng) * scale;
                if (rl && f < 0.f) f = 0.f;
                const amx::bf16_t hi = amx::f2bf_host(f);
                unsigned hu = (unsigned)hi << 16;
                float    hf;
                memcpy(&hf, &hu, 4);
                v[r * rs + k]       = hi;
                v[r * rs + p.K + k] = amx::f2bf_host(f - hf);
            }
    };
    fill(w, p.Npad, 0.02f, false);
    fill(x, p.Tpad, 1.f, relu);
    CK(hipMalloc((void**)&p.W, w.size() * 2));
    CK(hipMalloc((void**)&p.X, x.size() * 2));
    CK(hipMemcpy(p.W, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(p.X, x.data(), x.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&p.bias, p.Npad * 4));
    CK(hipMemset(p.bias, 0, p.Npad * 4));
    CK(hipMalloc((void**)&p.out, (size_t)p.T * p.N * 4));
    CK(hipMalloc((void**)&p.pmin, (size_t)(p.Npad / 128) * p.Tpad * 4));
    CK(hipMalloc((void**)&p.pidx, (size_t)(p.Npad / 128) * p.Tpad * 4));
    const size_t trace_n = (size_t)2048 * 64 * 4;
    CK(hipMalloc((void**)&p.trace, trace_n * 8));
    CK(hipMemset(p.trace, 0, trace_n * 8));
    CK(hipMalloc((void**)&p.sync, 8 * 64 * 4));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(amx::g_gemm_trace), &p.trace, sizeof(void*)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(amx::g_gemm_sync), &p.sync, sizeof(void*)));
    using CfgC  = amx::GemmCfg<256, 256, 2, 4, 2>;
    using XfgC  = amx::GemmCfg<256, 256, 2, 4, 2, 32, true>;
    using XfgA  = amx::GemmCfg<128, 128, 2, 2, 2, 32, true>;
    using CfgS  = amx::GemmCfg<128, 64, 2, 2, 2>;
    using CfgS3 = amx::GemmCfg<128, 64, 2, 2, 3>;
    using XfgS  = amx::GemmCfg<128, 64, 2, 2, 2, 32, true>;
    using XfgS3 = amx::GemmCfg<128, 64, 2, 2, 3, 32, true>;
    const int iters = getenv("PROBE_ITERS") ? atoi(getenv("PROBE_ITERS")) : 10;
    { Problem q = p; fprintf(stderr, "warm-up\n"); for (int w = 0; w < 30; ++w) run<CfgC, 128>(q, 10, true); }
    for (int a = 1; a < argc; ++a) {
        std::string s = argv[a];
        p.gt = p.gn = 0;
        if (s.find(':') != std::string::npos) {
            sscanf(s.c_str() + s.find(':') + 1, "%dx%d", &p.gt, &p.gn);
            s = s.substr(0, s.find(':'));
        }
        // pipelined 256x256 kernel: p<DBG> (bf16), xp<DBG> (split bf16)
        if (s == "p0") run_pipe<CfgC, 0>(p, iters);
        else if (s == "p2") run_pipe<CfgC, 2>(p, iters);
        else if (s == "p4") run_pipe<CfgC, 4>(p, iters);
        else if (s == "p8") run_pipe<CfgC, 8>(p, iters);          // no MFMA
        else if (s == "p16") run_pipe<CfgC, 16>(p, iters);        // no operand DMA after the first K-tile
        else if (s == "p32") run_pipe<CfgC, 32>(p, iters);        // DMA burst behind the barrier
        else if (s == "ptrace") { run_pipe<CfgC, 1>(p, 1); dump_trace(p, trace_n); }
        else if (s == "xp0") run_pipe<XfgC, 0>(p, iters);
        else if (s == "xp4") run_pipe<XfgC, 4>(p, iters);
        else if (s == "xp8") run_pipe<XfgC, 8>(p, iters);
        else if (s == "xp16") run_pipe<XfgC, 16>(p, iters);
        else if (s == "xp24") run_pipe<XfgC, 24>(p, iters);       // fragment reads + barriers only
        else if (s == "xp32") run_pipe<XfgC, 32>(p, iters);
        else if (s == "xp40") run_pipe<XfgC, 40>(p, iters);       // DMA burst, no MFMA
        else if (s == "xp72") run_pipe<XfgC, 72>(p, iters);       // no MFMA, L2-resident operands
        else if (s == "xp64") run_pipe<XfgC, 64>(p, iters);       // L2-resident operands
        else if (s == "p72") run_pipe<CfgC, 72>(p, iters);
        else if (s == "p64") run_pipe<CfgC, 64>(p, iters);
        else if (s == "xptrace") { run_pipe<XfgC, 1>(p, 1); dump_trace(p, trace_n); }
        else if (s == "xptrace32") { run_pipe<XfgC, 33>(p, 1); dump_trace(p, trace_n); }
        else if (s == "xptrace64") { run_pipe<XfgC, 65>(p, 1); dump_trace(p, trace_n); }
        // generic kernel, 256x256 tiles
        else if (s == "0") run<CfgC, 0>(p, iters);
        else if (s == "x0") run<XfgC, 0>(p, iters);
        else if (s == "xa0") run<XfgA, 0>(p, iters);
        else if (s == "64") run<CfgC, 64 | 128>(p, iters);       // operand streaming only, no epilogue
        else if (s == "128") run<CfgC, 128>(p, iters);           // K-loop only
        else if (s == "256") run<CfgC, 256>(p, iters);           // epilogue without global stores
        else if (s == "136") run<CfgC, 8 | 128>(p, iters);        // loads + fragment reads, no MFMA, no epilogue
        else if (s == "144") run<CfgC, 16 | 128>(p, iters);       // no global loads after the first: reads + MFMA
        // small tiles (PROBE_N=2048 PROBE_T=1024: a hidden layer of config 4)
        else if (s == "s0") run<CfgS, 0>(p, iters);
        else if (s == "s128") run<CfgS, 128>(p, iters);
        else if (s == "s192") run<CfgS, 64 | 128>(p, iters);
        else if (s == "t0") run<CfgS3, 0>(p, iters);
        else if (s == "t128") run<CfgS3, 128>(p, iters);
        else if (s == "xs0") run<XfgS, 0>(p, iters);
        else if (s == "xt0") run<XfgS3, 0>(p, iters);
        else if (s == "trace") { run<CfgC, 512>(p, 1); dump_trace(p, trace_n); }
        else fprintf(stderr, "unknown variant %s\n", s.c_str());
    }
    return 0;
}
