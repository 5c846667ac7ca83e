// tools/mx_probe.hip -- facts the round-4 NN scheme (f16 hi.hi + two MX cross products) and the fused GMM kernel's wave
// specialisation depend on, measured on the box:
//   A. operand model of v_mfma_scale_f32_32x32x64_f8f6f4 as the kernels use it: lane l holds row l & 31 and the 32 k of K-block
//      l >> 5 (fp4: 8 per register, low nibble first; fp8: 4 per register), ONE e8m0 scale per lane picked from a byte of the
//      scale register by op_sel; result in the common 32x32 accumulator layout.  Checked against a host evaluation.
//   B. matrix-pipe rates on random operands (the chip is power managed): f16 32x32x16 alone, MX fp8 / fp6 / fp4 32x32x64 alone,
//      and the instruction mixes of one K = 64 slab of the candidate schemes (4 f16 + 2 MX), as time per slab.
//   C. a matrix-only wave and a vector-only wave co-resident on every SIMD (MI355X_MICROARCH.md, wave scheduling): each alone and
//      both together.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mx_probe.hip -o tools/build/mx_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef int      v8i __attribute__((ext_vector_type(8)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------- A: operand model
template<int FMT>  // 0 fp8 e4m3, 4 fp4 e2m1
__global__ void mx_once(const v8i* a, const v8i* b, const unsigned* sa, const unsigned* sb, f32x16* c, int sel) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r)
        acc[r] = 0.f;
    const v8i av = a[threadIdx.x], bv = b[threadIdx.x];
    const int s1 = (int)sa[threadIdx.x], s2 = (int)sb[threadIdx.x];
    switch (sel) {
        case 0: acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, FMT, FMT, 0, s1, 0, s2); break;
        case 1: acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, FMT, FMT, 1, s1, 1, s2); break;
        case 2: acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, FMT, FMT, 2, s1, 2, s2); break;
        default: acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, FMT, FMT, 3, s1, 3, s2); break;
    }
    c[threadIdx.x] = acc;
}

static float fp4_value(unsigned code) {
    static const float mag[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    return (code & 8) ? -mag[code & 7] : mag[code & 7];
}
static float e4m3_value(unsigned code) {
    const int   e = (code >> 3) & 15, m = code & 7;
    const float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
    return (code & 0x80) ? -v : v;
}

static bool check_layout(int fmt) {
    std::vector<unsigned> ha(64 * 8), hb(64 * 8), hsa(64), hsb(64);
    std::vector<float>    A(32 * 64), B(32 * 64), SA(64), SB(64);
    unsigned              s = 12345u + fmt;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    bool all_ok = true;
    for (int sel = 0; sel < 4; ++sel) {
        for (int l = 0; l < 64; ++l) {
            const int row = l & 31, kb = l >> 5;
            for (int v = 0; v < 8; ++v) {
                unsigned wa = 0, wb = 0;
                if (fmt == 4) {
                    for (int n = 0; n < 8; ++n) {
                        const unsigned ca = rnd() & 15, cb = rnd() & 15;
                        wa |= ca << (4 * n);
                        wb |= cb << (4 * n);
                        if (v < 4) {
                            A[row * 64 + 32 * kb + 8 * v + n] = fp4_value(ca);
                            B[row * 64 + 32 * kb + 8 * v + n] = fp4_value(cb);
                        }
                    }
                }
                else {
                    for (int n = 0; n < 4; ++n) {
                        unsigned ca = rnd() & 255, cb = rnd() & 255;
                        if ((ca & 0x7f) == 0x7f) ca &= ~1u;  // no NaN
                        if ((cb & 0x7f) == 0x7f) cb &= ~1u;
                        wa |= ca << (8 * n);
                        wb |= cb << (8 * n);
                        A[row * 64 + 32 * kb + 4 * v + n] = e4m3_value(ca);
                        B[row * 64 + 32 * kb + 4 * v + n] = e4m3_value(cb);
                    }
                }
                ha[l * 8 + v] = wa;
                hb[l * 8 + v] = wb;
            }
            const unsigned ea = 120 + rnd() % 15, eb = 120 + rnd() % 15;
            SA[l]  = ldexpf(1.f, (int)ea - 127);
            SB[l]  = ldexpf(1.f, (int)eb - 127);
            hsa[l] = (rnd() & 0xffffff) << 8 | 0;  // garbage in the other bytes
            hsb[l] = (rnd() & 0xffffff) << 8 | 0;
            hsa[l] = (hsa[l] << 0);
            // place the scale in byte `sel`
            unsigned ga = rnd(), gb = rnd();
            ga = (ga & ~(0xffu << (8 * sel))) | (ea << (8 * sel));
            gb = (gb & ~(0xffu << (8 * sel))) | (eb << (8 * sel));
            hsa[l] = ga;
            hsb[l] = gb;
        }
        v8i *     da, *db;
        unsigned *dsa, *dsb;
        f32x16*   dc;
        CK(hipMalloc((void**)&da, 64 * 32));
        CK(hipMalloc((void**)&db, 64 * 32));
        CK(hipMalloc((void**)&dsa, 256));
        CK(hipMalloc((void**)&dsb, 256));
        CK(hipMalloc((void**)&dc, 64 * 64));
        CK(hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
        if (fmt == 4)
            hipLaunchKernelGGL(mx_once<4>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, sel);
        else
            hipLaunchKernelGGL(mx_once<0>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc, sel);
        std::vector<float> hc(64 * 16);
        CK(hipMemcpy(hc.data(), dc, 64 * 64, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double    ref = 0;
                for (int k = 0; k < 64; ++k)
                    ref += (double)A[i * 64 + k] * SA[i + 32 * (k >> 5)] * (double)B[j * 64 + k] * SB[j + 32 * (k >> 5)];
                worst = std::max(worst, std::fabs(ref - hc[l * 16 + r]) / (1.0 + std::fabs(ref)));
            }
        printf("  \"layout_fmt%d_opsel%d_worst_rel\": %.3g,\n", fmt, sel, worst);
        all_ok = all_ok && worst < 1e-5;
        hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dc);
    }
    return all_ok;
}

// ---------------------------------------------------------------------------------------------- B: rates
// MIX: 0 f16 only (4 per slab), 1 MX only (2 per slab, format FMT), 2 four f16 + two MX (one K = 64 slab of a 32x32 block)
template<int MIX, int FMT>
__global__ __launch_bounds__(256) void slab_rate(const unsigned* __restrict__ seed, float* __restrict__ sink, int iters) {
    unsigned s = seed[threadIdx.x & 63] + threadIdx.x;
    v8i      qa, qb;
    f16x8    ha, hb;
    for (int i = 0; i < 8; ++i) {
        s     = s * 1664525u + 1013904223u;
        qa[i] = (int)(s & 0x7f7f7f7fu & ~0x40404040u);  // fp8: small positive values (no NaN / inf); fp4 / fp6: any bits
        s     = s * 1664525u + 1013904223u;
        qb[i] = (int)(s & 0x7f7f7f7fu & ~0x40404040u);
        s     = s * 1664525u + 1013904223u;
        ha[i] = __builtin_bit_cast(_Float16, (unsigned short)((s >> 16) & 0x3bff));
        s     = s * 1664525u + 1013904223u;
        hb[i] = __builtin_bit_cast(_Float16, (unsigned short)((s >> 16) & 0x3bff));
    }
    if (FMT == 4 || FMT == 2) {
        for (int i = 0; i < 8; ++i) {
            s = s * 1664525u + 1013904223u; qa[i] = (int)s;
            s = s * 1664525u + 1013904223u; qb[i] = (int)s;
        }
    }
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
        for (int r = 0; r < 16; ++r)
            acc[k][r] = 0.f;
    const int sc = 0x7f7f7f7f - 0x04040404;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // four independent 32x32 blocks, one slab each
            if (MIX != 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[k], 0, 0, 0);
            }
            if (MIX != 0) {
                acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[k], FMT, FMT, 0, sc, 0, sc);
                acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qb, qa, acc[k], FMT, FMT, 0, sc, 0, sc);
            }
        }
    }
    float t = 0.f;
    for (int k = 0; k < 4; ++k)
        for (int r = 0; r < 16; ++r)
            t += acc[k][r];
    if (t == 123.456f)
        sink[threadIdx.x] = t;
}

// ---------------------------------------------------------------------------------------------- C: co-resident waves
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run a chain of f16 MFMAs, waves 4-7 the GMM distance step
// (v_sub, v_mul, v_mul, v_add on four chains).  which: 1 matrix waves only, 2 vector waves only, 3 both.
__global__ __launch_bounds__(512) void coresident(float* __restrict__ sink, int n_mfma, int n_valu, int which, float seed) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(which & 1))
            return;
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = (_Float16)(0.001f * (threadIdx.x & 7) + i + seed);
            b[i] = (_Float16)(0.5f - 0.01f * i);
        }
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k)
            for (int r = 0; r < 16; ++r)
                acc[k][r] = 0.f;
        for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
        }
        float t = 0.f;
        for (int k = 0; k < 4; ++k)
            for (int r = 0; r < 16; ++r)
                t += acc[k][r];
        if (t == 123.456f)
            sink[threadIdx.x] = t;
    }
    else {
        if (!(which & 2))
            return;
        float x0 = seed + threadIdx.x, x1 = x0 * 1.5f, x2 = x0 + 2.f, x3 = x0 - 3.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const float m = 0.25f, isr = 1.0001f;
        for (int it = 0; it < n_valu; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %0, %9, %0\n v_mul_f32 %4, %0, %0\n v_add_f32 %4, %4, %0\n"
                             "v_sub_f32 %1, %1, %8\n v_mul_f32 %1, %9, %1\n v_mul_f32 %5, %1, %1\n v_add_f32 %5, %5, %1\n"
                             "v_sub_f32 %2, %2, %8\n v_mul_f32 %2, %9, %2\n v_mul_f32 %6, %2, %2\n v_add_f32 %6, %6, %2\n"
                             "v_sub_f32 %3, %3, %8\n v_mul_f32 %3, %9, %3\n v_mul_f32 %7, %3, %3\n v_add_f32 %7, %7, %3"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                             : "v"(m), "v"(isr));
            }
        }
        const float t = x0 + x1 + x2 + x3 + s0 + s1 + s2 + s3;
        if (t == 123.456f)
            sink[threadIdx.x] = t;
    }
}

static float time_ms(hipEvent_t a, hipEvent_t b) {
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("{\n");
    const bool ok4 = check_layout(4), ok8 = check_layout(0);
    printf("  \"operand_model_holds_fp4\": %s, \"operand_model_holds_fp8\": %s,\n", ok4 ? "true" : "false", ok8 ? "true" : "false");
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned* seed;
    float*    sink;
    CK(hipMalloc((void**)&seed, 64 * 4));
    CK(hipMalloc((void**)&sink, 4096));
    std::vector<unsigned> hs(64);
    for (int i = 0; i < 64; ++i)
        hs[i] = 12345u + 977u * i;
    CK(hipMemcpy(seed, hs.data(), 256, hipMemcpyHostToDevice));
    const int iters = 4000, blocks = p.multiProcessorCount * 2;  // 2 waves per SIMD
    auto run = [&](const char* name, void (*k)(const unsigned*, float*, int), double units_per_slab) {
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, seed, sink, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            if (rep)
                best = std::min(best, (double)time_ms(e0, e1));
        }
        const double slabs = (double)blocks * 4 * iters * 4;  // waves x iterations x blocks per iteration
        const double ns_per_slab_simd = best * 1e6 / ((double)iters * 4 * 2);  // two waves share a SIMD
        // "algorithmic" rate: one slab = 2*32*32*64 useful flops of the f32 product it stands for
        printf("  \"%s\": {\"ms\": %.3f, \"ns_per_slab_per_simd\": %.2f, \"algorithmic_TFLOPs\": %.0f, \"datasheet_units\": %.2f},\n", name, best,
               ns_per_slab_simd, slabs * 2.0 * 32 * 32 * 64 / (best * 1e-3) / 1e12, units_per_slab);
    };
    run("slab_f16_only (4 x 32x32x16 f16: 1 unit)", slab_rate<0, 0>, 1.0);
    run("slab_fp8_only (2 x 32x32x64 e4m3)", slab_rate<1, 0>, 1.0);
    run("slab_fp6_only (2 x 32x32x64 e2m3)", slab_rate<1, 2>, 0.5);
    run("slab_fp4_only (2 x 32x32x64 e2m1)", slab_rate<1, 4>, 0.5);
    run("slab_f16_plus_2_fp8 (2 units)", slab_rate<2, 0>, 2.0);
    run("slab_f16_plus_2_fp6 (1.5 units)", slab_rate<2, 2>, 1.5);
    run("slab_f16_plus_2_fp4 (1.5 units)", slab_rate<2, 4>, 1.5);
    // C
    {
        const int nm = 20000, nv = 20000 * 4 * 32 / (128 * 5);  // ~ equal time alone: 4 MFMAs x 32 cycles vs 128 VALU x ~5 ticks
        double    t[4] = {0, 0, 0, 0};
        for (int which = 1; which <= 3; ++which) {
            double best = 1e30;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(coresident, dim3(p.multiProcessorCount), dim3(512), 0, 0, sink, nm, nv, which, 1.0f);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                if (rep)
                    best = std::min(best, (double)time_ms(e0, e1));
            }
            t[which] = best;
        }
        printf("  \"coresident\": {\"mfma_wave_alone_ms\": %.3f, \"valu_wave_alone_ms\": %.3f, \"both_ms\": %.3f, \"sum_ms\": %.3f, \"max_ms\": %.3f, "
               "\"overlap_fraction\": %.3f},\n",
               t[1], t[2], t[3], t[1] + t[2], std::max(t[1], t[2]), (t[1] + t[2] - t[3]) / std::min(t[1], t[2]));
    }
    printf("  \"device\": \"%s\", \"compute_units\": %d\n}\n", p.name, p.multiProcessorCount);
    return 0;
}
