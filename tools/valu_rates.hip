// tools/valu_rates.hip -- issue cost (cycles per wave64 instruction on one SIMD) of the VALU / DS instructions the GMM kernels are
// made of, measured with s_memtime around long unrolled chains.  One wave per SIMD (256 threads, 1 block per CU) and 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/build/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template<int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 * 1.5f, a2 = a0 + 2.f, a3 = a0 - 3.f, a4 = a1 + 1.f, a5 = a2 * 0.5f, a6 = a3 + 7.f, a7 = a4 - 9.f;
    float b0 = 1.0001f, b1 = 0.9999f;
    double d0 = a0, d1 = a1;
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 ^ 0x5bd1e995u;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    f32x16 acc0, acc1;
    f16x8  fa, fb;
    for (int i = 0; i < 16; ++i) { acc0[i] = a0 + i; acc1[i] = a1 - i; }
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.001f * (threadIdx.x & 7) + i); fb[i] = (_Float16)(0.5f - 0.01f * i); }
    __builtin_amdgcn_s_barrier();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 256; ++it) {
        if (OP == 0) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 1) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&b0)); asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));) }
        if (OP == 2) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 3) { REP64(asm volatile("v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %1, %1, %2, 31\n v_alignbit_b32 %2, %2, %3, 31\n v_alignbit_b32 %3, %3, %0, 31" : "+v"(u0), "+v"(u1), "+v"(*(unsigned*)&a2), "+v"(*(unsigned*)&a3));) }
        if (OP == 4) { REP64(asm volatile("v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_min3_f32 %2, %2, %3, %0\n v_min3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 5) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
        if (OP == 6) { REP64(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %0" : "+v"(d0), "+v"(d1));) }
        if (OP == 7) { REP64(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&b0)); asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));) }
        if (OP == 8) { REP64(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 9) { REP64(asm volatile("v_cvt_f64_f32 %0, %2\n v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %3\n v_cvt_f32_f64 %1, %0" : "+v"(d0), "+v"(a1) : "v"(a2), "v"(a3));) }
        if (OP == 10) { REP64(asm volatile("ds_bpermute_b32 %0, %2, %0\n ds_bpermute_b32 %1, %2, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1) : "v"(u0 & 252));) }
        if (OP == 11) { REP64(asm volatile("v_cmp_gt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %1, %0" :: "v"(d0), "v"(d1) : "vcc");) }
        if (OP == 12) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n v_pk_fma_f32 %1, %1, %2, %1" : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&b0)); asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n v_pk_fma_f32 %1, %1, %2, %1" : "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));) }
        if (OP == 13) { REP64(asm volatile("v_mul_f32 %0, s0, %0\n v_mul_f32 %1, s1, %1\n v_mul_f32 %2, s2, %2\n v_mul_f32 %3, s3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 14) { REP64(asm volatile("v_pk_mul_f32 %0, s[0:1], %0\n v_pk_mul_f32 %1, s[2:3], %1" : "+v"(*(double*)&a0), "+v"(*(double*)&a2)); asm volatile("v_pk_mul_f32 %0, s[0:1], %0\n v_pk_mul_f32 %1, s[2:3], %1" : "+v"(*(double*)&a4), "+v"(*(double*)&a6));) }
        if (OP == 16) { REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[4:5]\n v_cndmask_b32_e64 %1, %1, %2, s[4:5]\n v_cndmask_b32_e64 %2, %2, %3, s[4:5]\n v_cndmask_b32_e64 %3, %3, %0, s[4:5]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 17) { REP64(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %0" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
        if (OP == 18) { REP64(asm volatile("v_cmp_gt_f32_e64 s[4:5], %0, %1\n v_cmp_gt_f32_e64 s[6:7], %1, %2\n v_cmp_gt_f32_e64 s[8:9], %2, %3\n v_cmp_gt_f32_e64 s[10:11], %3, %0" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s4","s5","s6","s7","s8","s9","s10","s11");) }
        if (OP == 19) { REP64(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
        if (OP == 20) { REP64(asm volatile("v_bfi_b32 %0, %0, %1, %2\n v_bfi_b32 %1, %1, %2, %3\n v_bfi_b32 %2, %2, %3, %0\n v_bfi_b32 %3, %3, %0, %1" : "+v"(u0), "+v"(u1), "+v"(*(unsigned*)&a2), "+v"(*(unsigned*)&a3));) }
        if (OP == 21) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 22) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n v_mul_f32 %5, %5, %4\n v_mul_f32 %6, %6, %4\n v_mul_f32 %7, %7, %4\n v_mul_f32 %8, %8, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(a4), "v"(a5), "v"(a6), "v"(a7));) }
        if (OP == 23) { REP64(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 24) { REP64(asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %1, %2, vcc\n v_cndmask_b32 %0, %2, %1, vcc\n v_cndmask_b32 %3, %2, %1, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");) }
        if (OP == 25) { REP64(asm volatile("v_mul_f32_dpp %0, %4, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %4, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %2, %4, %2 row_newbcast:11 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %4, %3 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 26) { REP64(asm volatile("v_sub_f32 %0, %0, %4\n v_mul_f32_dpp %0, %4, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mul_f32 %1, %0, %0\n v_add_f32 %2, %2, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 27) { REP64(asm volatile("v_sub_f32 %0, %0, %4\n v_mul_f32 %0, s0, %0\n v_mul_f32 %1, %0, %0\n v_add_f32 %2, %2, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 28) { REP64(asm volatile("v_cmp_lt_f32 vcc, %2, %3\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_cmp_lt_f32 vcc, %3, %2\n v_addc_co_u32 %1, vcc, %1, %1, vcc" : "+v"(u0), "+v"(u1) : "v"(a2), "v"(a3) : "vcc");) }
        if (OP == 29) { REP64(asm volatile("v_sub_f32 %4, %2, %3\n v_alignbit_b32 %0, %0, %4, 31\n v_sub_f32 %4, %3, %2\n v_alignbit_b32 %1, %1, %4, 31" : "+v"(u0), "+v"(u1) : "v"(a2), "v"(a3), "v"(a4));) }
        if (OP == 30) { REP64(asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb));) }
        if (OP == 31) { REP64(asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n v_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(acc0), "+v"(acc1) : "v"(fa), "v"(fb), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0));) }
        if (OP == 32) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (OP == 15) { REP64(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)d0 + (float)d1 + (float)(u0 + u1) + acc0[3] + acc1[5];
    if (r == 123.456f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template<int OP>
void run(const char* name, int per_rep, unsigned long long* d) {
    // wall-clock view: total wave-instructions / (kernel time x 1024 SIMDs); grid sized for 1, 2, 4 and 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int cfg[4][2] = {{256, 256}, {256, 512}, {256, 1024}, {512, 1024}};
    printf("%-30s", name);
    for (int c = 0; c < 4; ++c) {
        const int blocks = cfg[c][0], threads = cfg[c][1];
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h = 0;
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        const double n = 256.0 * 64 * per_rep;                    // instructions per wave
        const double waves_per_simd = (double)blocks * (threads / 64) / 1024.0;
        const double ns_per_simd_instr = ms * 1e6 / (n * waves_per_simd);
        printf(" | %g w/SIMD: %.2f ns/SIMD-instr, %.2f ticks/wave-instr", waves_per_simd, ns_per_simd_instr, h / n);
    }
    printf("\n");
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 16);
    run<0>("v_mul_f32", 4, d);
    run<8>("v_add_f32", 4, d);
    run<2>("v_fma_f32", 4, d);
    run<13>("v_mul_f32 (sgpr operand)", 4, d);
    run<25>("v_mul_f32_dpp row_newbcast", 4, d);
    run<26>("distance step, isr by DPP", 4, d);
    run<27>("distance step, isr in SGPR", 4, d);
    run<28>("v_cmp + v_addc (mask shift-in)", 4, d);
    run<30>("v_mfma_f32_32x32x16_f16 alone (2 chains)", 2, d);
    run<32>("16 plain VALU alone", 16, d);
    run<31>("2 MFMA + 16 plain VALU interleaved", 18, d);
    run<29>("v_sub + v_alignbit (mask)", 4, d);
    run<1>("v_pk_mul_f32", 4, d);
    run<14>("v_pk_mul_f32 (sgpr pair)", 4, d);
    run<7>("v_pk_add_f32", 4, d);
    run<12>("v_pk_fma_f32", 4, d);
    run<3>("v_alignbit_b32", 4, d);
    run<4>("v_min3_f32", 4, d);
    run<5>("v_cndmask_b32", 4, d);
    run<6>("v_add_f64", 2, d);
    run<9>("v_cvt f64<->f32", 4, d);
    run<11>("v_cmp_gt_f64", 2, d);
    run<16>("v_cndmask_b32_e64 sgpr", 4, d);
    run<24>("v_cndmask_b32 vcc (no chain)", 4, d);
    run<17>("v_cmp_gt_f32 vcc", 4, d);
    run<18>("v_cmp_gt_f32_e64 sgpr", 4, d);
    run<19>("v_cmp + v_cndmask pairs", 4, d);
    run<20>("v_bfi_b32", 4, d);
    run<21>("v_mul_f32 dependent chain", 4, d);
    run<22>("v_mul_f32 8 chains", 8, d);
    run<23>("v_mov_b32", 4, d);
    run<15>("v_permlane32_swap", 2, d);
    run<10>("ds_bpermute_b32 (+wait)", 2, d);
    return 0;
}
