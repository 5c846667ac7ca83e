// gmm_fused.hip -- ONE kernel for the screened maximum-approximation GMM scorer (pooled covariance, dim <= 40, <= 16 private
// densities per mixture): f16 MFMA screen, survivor selection and the exact f32 / f64 evaluation of the survivors
// (Mm/GaussDiagonalMaximumFeatureScorer.cc:116-180) without the survivor masks ever leaving the registers.
//
// Round 1 ran this as two kernels (gmm_screen_rows_kernel -> 2 B of mask per frame and mixture slot in HBM ->
// gmm_screen_exact_kernel with thread = frame, 79 KB of LDS and 244 VGPRs at two waves per SIMD; 1.8 + 4.5 ms per 63 936 frames,
// the feature rows re-read once per 16-mixture tile).  Here a wavefront owns 32 frames for the whole pass and walks the
// model's 16-mixture tiles:
//   * the tile arrives as ONE contiguous record (f16 screen rows, f32 mean rows with the f64 constant in the row padding,
//     per-mixture threshold terms and density counts) by LDS-DMA into one of two LDS stages (2 x 77 KB), one s_barrier per tile;
//     the stores of a tile are issued behind the next tile's DMA, so the only wait at the top of a tile is
//     s_waitcnt vmcnt(#stores) -- it waits for the DMA, not for the stores;
//   * screen: 8 blocks of 4 v_mfma_f32_32x32x16_f16 (frame fragments stay in registers); in the 32x32 accumulator layout lane
//     (frame, half h) holds the 16 slots of mixture 2b + h of block b, so minimum, threshold and the 16-bit survivor mask need
//     no cross-lane traffic (identical arithmetic to gmm_screen_rows_kernel: same masks);
//   * exact stage in the SAME lane layout: a lane owns 8 (frame, mixture) pairs per tile.  Every pair has a first survivor
//     (the screen keeps the minimum), so the 8 first survivors are evaluated in lockstep -- static register indices, the next
//     mean row (ds_read_b128, rows 16 B-slot staggered) fetched during the current distance -- and the ~4 % further survivors
//     follow in a short divergent loop, in slot order, through select chains on the 8 running (best, index) pairs;
//   * a frame's 16 scores / 16 best densities leave as 2 x 32 B per lane after ONE v_permlane32_swap per pair of values
//     (no LDS transposition), and the best state of a frame is carried across tiles in registers: the per-tile arg-min
//     partials of round 1 (320 MB per pass) shrink to one (min, state) pair per frame and workgroup.
// HBM traffic per pass = scores + best densities (8 B per frame and mixture) + the model records once per XCD.
#include "common.hpp"
#include "gmm_device.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <type_traits>
#include <vector>

namespace amx {

typedef _Float16 fus_f16x8 __attribute__((ext_vector_type(8)));
typedef float    fus_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned fus_u32x2 __attribute__((ext_vector_type(2)));

// mean rows: 16-byte aligned and an ODD number of 16-byte slots long (rows of one mixture start in 16 different slots:
// conflict-free ds_read_b128 when 16 lanes read 16 different rows); the last two floats of a row hold the density's f64 constant
__host__ __device__ constexpr int fused_ld(int dim) {
    return 4 * (((dim + 3) / 4) | 1);
}
// a tile record = [screen part: 256 f16 slot rows (32 KB) + p1[16], p2[16], nd[16], padded to 33 KB][mean part: 256 x LD f32]
constexpr int kFusedABytes = 256 * 128;
constexpr int kFusedAStage = kFusedABytes + 1024;
__host__ __device__ constexpr int fused_mu_stage(int dim) {
    return 256 * fused_ld(dim) * 4;
}
__host__ __device__ constexpr int fused_rec_bytes(int dim) {
    return kFusedAStage + fused_mu_stage(dim);
}

// 1 KB of a tile record per wave-instruction straight into LDS (global_load_lds_dwordx4: LDS address = M0 + lane * 16).
// Inline assembly on purpose: for an LDS-DMA it knows about, the compiler's wait-count pass puts s_waitcnt vmcnt(0) in front of
// the next LDS read it cannot prove disjoint -- here the mean-row reads of the CURRENT stage, i.e. every tile would wait for the
// NEXT tile's DMA (and for its own stores).  The ordering is explicit instead: counted vmcnt + s_barrier at the top of a tile.
// Nothing else in the kernel uses M0.
__device__ __forceinline__ void fused_dma16(const void* gsrc, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// Per wavefront and tile the kernel interleaves two things (software pipeline across tiles):
//   S(k+1)  the screen of the NEXT tile: per block 4 MFMAs + minimum / threshold / 16-bit survivor mask   (matrix pipe + VALU)
//   E(k)    the exact evaluation of THIS tile's first survivors, one per mixture, in lockstep              (VALU only)
// block by block, so that the MFMAs (and their dependent-issue latency) run underneath the distance arithmetic instead of in a
// phase of their own -- with the per-tile barrier all eight waves of a workgroup would otherwise be in the MFMA phase together
// and in the VALU phase together (measured: screen phase alone 1.8 of 5.8 ms, matrix pipe 35 % busy, VALU idle most of it).
// LDS: screen ring 2 x 33 KB + mean ring 2 x 44 KB; at the top of iteration k (one s_barrier) the DMA of mean part k+1 and
// screen part k+2 is issued into the two slots that iteration k-1 has just finished with.
template<int DIM, bool BEST>
__global__ __launch_bounds__(512, 2) void gmm_fused_kernel(const float* __restrict__ g_feats, const _Float16* __restrict__ g_X,
                                                          const float* __restrict__ g_nx, const float* __restrict__ g_q,
                                                          const char* __restrict__ g_rec, const float* __restrict__ g_isr,
                                                          float* __restrict__ g_scores, uint32_t* __restrict__ g_best, int T, int n_mix,
                                                          int n_tiles, int r_split, float* __restrict__ g_part_min,
                                                          unsigned* __restrict__ g_part_idx, int part_ld,
                                                          unsigned long long* __restrict__ g_survivors, int abl) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int LD = fused_ld(DIM), REC = fused_rec_bytes(DIM), MU_STAGE = fused_mu_stage(DIM);
    constexpr int NPA = kFusedAStage / 1024, NPM = MU_STAGE / 1024;
    constexpr int MU_RING = 2 * kFusedAStage;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_t = blockIdx.x / r_split, part = blockIdx.x % r_split;
    const int per = (n_tiles + r_split - 1) / r_split;
    const int r_begin = part * per, r_end = min(n_tiles, r_begin + per);
    if (r_begin >= r_end)
        return;
    const int      n_it = r_end - r_begin;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    auto load_A = [&](int r, int slot) {
        const char*    src = g_rec + (size_t)r * REC + lane * 16;
        const unsigned dst = lds_base + slot * kFusedAStage;
        for (int p = wave; p < NPA; p += 8)
            fused_dma16(src + p * 1024, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + p * 1024)));
    };
    auto load_mu = [&](int r, int slot) {
        const char*    src = g_rec + (size_t)r * REC + kFusedAStage + lane * 16;
        const unsigned dst = lds_base + MU_RING + slot * MU_STAGE;
        for (int p = wave; p < NPM; p += 8)
            fused_dma16(src + p * 1024, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + p * 1024)));
    };
    load_A(r_begin, 0);
    load_mu(r_begin, 0);
    if (n_it > 1)
        load_A(r_begin + 1, 1);
    const int  frow = lane & 31, fk = lane >> 5;
    const int  t    = tile_t * 256 + wave * 32 + frow;  // < Tpad: the packed operand rows exist
    const bool live = t < T;
    const int  tt   = live ? t : T - 1;
    const bool wave_live = tile_t * 256 + wave * 32 < T;  // wave-uniform
    float      x[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i)
        x[i] = g_feats[(size_t)tt * DIM + i];
    fus_f16x8 bx[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        bx[ks] = *(const fus_f16x8*)(g_X + (size_t)t * 64 + (ks * 2 + fk) * 8);
    const float nx = g_nx[t], q = g_q[t];
    const bool  all = !(nx < __builtin_inff());  // operand row did not fit f16: keep every slot
    // output side: lane L stores 16 bytes = mixtures 4 (L & 3) .. + 3 of frame 16 s + (L >> 2) of the wave (s = 0, 1), so that four
    // adjacent lanes cover the 64 contiguous bytes of one frame (one 64-byte request instead of four 16-byte ones)
    const int  oj = lane & 3;
    const int  ot0 = tile_t * 256 + wave * 32 + (lane >> 2), ot1 = ot0 + 16;
    const int  src0 = 4 * ((lane >> 2) + 32 * (oj >> 1)), src1 = src0 + 64;  // ds_bpermute byte addresses of the source lanes
    const bool ohi = (oj & 1) != 0;
    // stores of 16 bytes need aligned rows; otherwise (and on the model's last, partial tile) scalar guarded stores
    const bool wide_ok = (n_mix & 3) == 0 && ((uintptr_t)g_scores & 15) == 0 && (!BEST || ((uintptr_t)g_best & 15) == 0);
    const bool counted = wide_ok && wave_live;  // this wave issues exactly (BEST ? 4 : 2) stores per full tile
    // The compiler's wait-count pass does not see the inline-asm waits of the loop: without a wait it can see, it would put its own
    // vmcnt(0) in front of the first use of x[] inside the loop -- and there that waits for the next tile's DMA and the stores.
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_s_barrier();
    float    run_min = 3.402823466e+38f;  // best state of this lane's mixtures so far (ascending state, strict '<')
    unsigned run_idx = 0xffffffffu;
    unsigned n_surv  = 0;  // densities this lane evaluated exactly (bench: survivors per mixture)

    // ---- screen of one block (two mixtures: 2 i + fk for lane half fk): 4 MFMAs, then minimum / threshold / survivor mask
    auto screen_mfma = [&](const char* stageA, int i) {
        const int  rr = i * 32 + frow;
        fus_f32x16 c;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            c[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const fus_f16x8 a = *(const fus_f16x8*)(stageA + rr * 128 + (((ks * 2 + fk) ^ ((rr >> 1) & 7)) << 4));
            c                 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bx[ks], c, 0, 0, 0);
        }
        return c;
    };
    auto screen_mask = [&](const char* stageA, int i, const fus_f32x16& c) -> unsigned {
        const float* s_p = (const float*)(stageA + kFusedABytes);  // p1[16], p2[16], (int) densities per mixture [16]
        float        mn = min3_first(c[0], c[1], c[2]);
        mn              = min3_raw(mn, c[3], c[4]);
        mn              = min3_raw(mn, c[5], c[6]);
        mn              = min3_raw(mn, c[7], c[8]);
        mn              = min3_raw(mn, c[9], c[10]);
        mn              = min3_raw(mn, c[11], c[12]);
        mn              = min3_raw(mn, c[13], c[14]);
        mn              = min3_raw(mn, c[15], c[15]);
        // tau as in gmm_screen_epilogue (gmm.hip): p1 = 2.2e-3 na + 1.3e-4 sqrtK, p2 = 1.3e-4 sqrtK na + 1.6e-5 cabs, q per frame
        const float p1 = s_p[i * 2 + fk], p2 = s_p[16 + i * 2 + fk];
        const int   nd = ((const int*)s_p)[32 + i * 2 + fk];
        const float thr = mn + fmaf(nx, p1, fmaf(fabsf(mn), 1.6e-5f, p2 + q)) + 1e-30f;
        unsigned    bits = 0;  // bit k = "slot k is above the threshold", filled from the top down
#pragma unroll
        for (int e = 14; e >= 0; e -= 2) {
            const gmm_pk2 dd = gmm_pk2{thr, thr} - gmm_pk2{c[e], c[e + 1]};
            bits             = __builtin_amdgcn_alignbit(bits, __float_as_uint(dd.y), 31);  // (bits << 1) | sign(thr - g)
            bits             = __builtin_amdgcn_alignbit(bits, __float_as_uint(dd.x), 31);
        }
        const unsigned valid = (1u << nd) - 1u;  // nd <= 16
        // frames behind the last one have an all-zero operand row (every slot ties): no survivors for them
        return live ? ((all ? 0xffffu : (~bits & 0xffffu)) & valid) : 0u;
    };

    unsigned M[4] = {0, 0, 0, 0};  // survivor masks of this lane's 8 mixtures (2 i + fk) of the current tile, 16 bit each
    if (wave_live) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const fus_f32x16 c = screen_mfma(lds, i);
            M[i >> 1] |= screen_mask(lds, i, c) << (16 * (i & 1));
        }
    }

    // ---- one iteration: exact evaluation of tile r (masks M, mean slot k & 1), interleaved with the screen of tile r + 1
    auto tile_body = [&](auto next_tag, int k) {
        constexpr bool NEXT = decltype(next_tag)::value;
        const int      r = r_begin + k;
        const char*    stageA = lds + ((k + 1) & 1) * kFusedAStage;  // screen part of tile r + 1
        const float*   s_mu = (const float*)(lds + MU_RING + (k & 1) * MU_STAGE);
        auto           fetch = [&](float (&dst)[DIM], double& cc, int row) {
            const float* src = s_mu + row * LD;
#pragma unroll
            for (int i = 0; i + 3 < DIM; i += 4) {
                const float4 v = *(const float4*)(src + i);
                dst[i]         = v.x;
                dst[i + 1]     = v.y;
                dst[i + 2]     = v.z;
                dst[i + 3]     = v.w;
            }
#pragma unroll
            for (int i = DIM & ~3; i < DIM; ++i)
                dst[i] = src[i];
            cc = *(const double*)(src + LD - 2);
        };
        n_surv += __popc(M[0]) + __popc(M[1]) + __popc(M[2]) + __popc(M[3]);
        float    best[8];
        unsigned bidx[8];
        int      slot[8];
        unsigned R[4] = {0, 0, 0, 0};  // further survivors (first one removed)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned m16 = (M[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            slot[i]            = m16 ? __ffs((int)m16) - 1 : -1;
            R[i >> 1] |= (m16 & (m16 - 1u)) << (16 * (i & 1));
            best[i] = FLT_MAX;
            bidx[i] = 0xffffffffu;
        }
        auto eval = [&](const float (&mu)[DIM], double cc, int sl, float& b, unsigned& bi) {
            const float  dist = gmm_distance_pk_reg<DIM>(x, mu, g_isr);
            const double s    = cc + (double)dist;
            const bool   take = sl >= 0 && (double)b > s;  // reference: if (bestScore > score) with an f32 bestScore
            b                 = take ? (float)s : b;
            bi                = take ? (unsigned)sl : bi;
        };
        unsigned Mn[4] = {0, 0, 0, 0};
        {
            // first survivor of every mixture in lockstep (static register indices; the next mean row is fetched during the
            // current distance), block i of the next tile's screen around it: MFMAs in front, mask arithmetic behind
            float  mua[DIM], mub[DIM];
            double ca, cb;
            fetch(mua, ca, fk * 16 + max(slot[0], 0));
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                fus_f32x16 c0, c1;
                if (NEXT)
                    c0 = screen_mfma(stageA, i);
                fetch(mub, cb, ((i + 1) * 2 + fk) * 16 + max(slot[i + 1], 0));
                if (!(abl & 8))
                    eval(mua, ca, slot[i], best[i], bidx[i]);
                if (NEXT) {
                    Mn[i >> 1] |= screen_mask(stageA, i, c0);
                    c1 = screen_mfma(stageA, i + 1);
                }
                if (i + 2 < 8)
                    fetch(mua, ca, ((i + 2) * 2 + fk) * 16 + max(slot[i + 2], 0));
                if (!(abl & 8))
                    eval(mub, cb, slot[i + 1], best[i + 1], bidx[i + 1]);
                if (NEXT)
                    Mn[i >> 1] |= screen_mask(stageA, i + 1, c1) << 16;
            }
        }
        // the ~4 % further survivors, in slot order, through select chains on the 8 running (best, index) pairs
        while (!(abl & 4) && __any((R[0] | R[1] | R[2] | R[3]) != 0u)) {
            const unsigned w01 = R[0] ? R[0] : R[1], w23 = R[2] ? R[2] : R[3];
            const bool     lo  = (R[0] | R[1]) != 0u;
            const unsigned rw  = lo ? w01 : w23;
            if (rw) {
                const int      w   = lo ? (R[0] ? 0 : 1) : (R[2] ? 2 : 3);
                const int      pos = __ffs((int)rw) - 1;
                const unsigned cleared = rw & (rw - 1u);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    R[j] = (w == j) ? cleared : R[j];
                const int i = 2 * w + (pos >> 4), sl = pos & 15;
                float     mu[DIM];
                double    cc;
                fetch(mu, cc, (i * 2 + fk) * 16 + sl);
                float b = best[0];
#pragma unroll
                for (int j = 1; j < 8; ++j)
                    b = (i == j) ? best[j] : b;
                const float  dist = gmm_distance_pk_reg<DIM>(x, mu, g_isr);
                const double s    = cc + (double)dist;
                const bool   take = (double)b > s;
                const float  nb   = (float)s;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool hit = take && i == j;
                    best[j]        = hit ? nb : best[j];
                    bidx[j]        = hit ? (unsigned)sl : bidx[j];
                }
            }
        }

        // ---- results: score = 0.5 * best (f32 * double -> f32 in the reference: the same value), best density, best state so far
        const int m0 = r * 16;
        float     sc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sc[i]       = 0.5f * best[i];
            const int m = m0 + i * 2 + fk;
            if (g_part_min && m < n_mix && sc[i] < run_min) {
                run_min = sc[i];
                run_idx = (unsigned)m;
            }
        }
        // lane halves exchange four values each (v_permlane32_swap): half 0 then holds mixtures 0..7 of the tile, half 1 mixtures
        // 8..15, in order; a ds_bpermute pass turns that into 16 contiguous bytes per lane, four adjacent lanes per frame
        unsigned so[8], bo[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const fus_u32x2 rs = __builtin_amdgcn_permlane32_swap(__float_as_uint(sc[j]), __float_as_uint(sc[4 + j]), false, false);
            so[2 * j]          = rs.x;
            so[2 * j + 1]      = rs.y;
            if (BEST) {
                const fus_u32x2 rb = __builtin_amdgcn_permlane32_swap(bidx[j], bidx[4 + j], false, false);
                bo[2 * j]          = rb.x;
                bo[2 * j + 1]      = rb.y;
            }
        }
        if (!(abl & 1)) {
            const bool full = wide_ok && m0 + 16 <= n_mix;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int  st  = s2 ? ot1 : ot0;
                const int  src = s2 ? src1 : src0;
                const bool ok  = st < T;
                unsigned   v[4], vb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned a0 = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)so[e]);
                    const unsigned a1 = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)so[4 + e]);
                    v[e]              = ohi ? a1 : a0;
                    if (BEST) {
                        const unsigned b0 = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)bo[e]);
                        const unsigned b1 = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)bo[4 + e]);
                        vb[e]             = ohi ? b1 : b0;
                    }
                }
                float*    gs = g_scores + (size_t)st * n_mix + m0 + oj * 4;
                uint32_t* gb = BEST ? g_best + (size_t)st * n_mix + m0 + oj * 4 : nullptr;
                if (full) {
                    if (ok) {
                        *(uint4*)gs = make_uint4(v[0], v[1], v[2], v[3]);
                        if (BEST)
                            *(uint4*)gb = make_uint4(vb[0], vb[1], vb[2], vb[3]);
                    }
                }
                else if (ok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m0 + oj * 4 + e < n_mix) {
                            gs[e] = __uint_as_float(v[e]);
                            if (BEST)
                                gb[e] = vb[e];
                        }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            M[j] = Mn[j];
    };

    for (int k = 0; k < n_it; ++k) {
        // my DMA pieces for this iteration are older than the stores of the previous one: waiting until only those stores are
        // outstanding means the pieces have landed (gfx9 retires vector memory operations in issue order)
        if (k == 0 || !counted)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (BEST)
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everybody's pieces are there, and nobody reads the slots refilled below any more
        if (!(abl & 2)) {
            if (k + 1 < n_it)
                load_mu(r_begin + k + 1, (k + 1) & 1);
            if (k + 2 < n_it)
                load_A(r_begin + k + 2, k & 1);
        }
        if (!wave_live)  // a wave behind the last frame only takes part in the DMA and the barriers
            continue;
        if (k + 1 < n_it)
            tile_body(std::true_type{}, k);
        else
            tile_body(std::false_type{}, k);
    }
    if (g_survivors) {
        unsigned long long n = live ? n_surv : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            n += __shfl_xor(n, off, 64);
        if (lane == 0 && n)
            atomicAdd(g_survivors, n);
    }
    if (g_part_min) {  // this workgroup's (min, state) of every frame: the partner lane holds the other half of the states
        const float    om = __shfl_xor(run_min, 32, 64);
        const unsigned oi = (unsigned)__shfl_xor((int)run_idx, 32, 64);
        if (om < run_min || (om == run_min && oi < run_idx)) {
            run_min = om;
            run_idx = oi;
        }
        if (fk == 0 && live) {
            g_part_min[(size_t)part * part_ld + t] = run_min;
            g_part_idx[(size_t)part * part_ld + t] = run_idx;
        }
    }
}

}  // namespace amx

// ---- host side (internal to librasr_amd.so; called from gmm.hip)

extern "C" int amx_internal_gmm_fused_supported(int dim, int pooled, int Kp) {
    if (!pooled || Kp != 64)
        return 0;
    switch (dim) {
        case 16: case 24: case 32: case 33: case 39: case 40: return 1;
        default: return 0;
    }
}

// Tile records [n_tiles][fused_rec_bytes(dim)]:
//   [0, 32768)            the tile's 256 f16 screen rows in gmm_screen_rows_kernel's row order, 16-byte chunks already XOR-swizzled
//                         (chunk c of row r sits at position c ^ ((r >> 1) & 7)) so that the LDS-DMA is a linear copy
//   [32768, +192)         p1[16], p2[16] (threshold terms per mixture), int nd[16] (densities per mixture); padded to 33 KB
//   [33792, +256 LD 4)    f32 mean rows, row = mixture * 16 + slot, the density's (f64) m2lw + logNorm in the last two floats
extern "C" int amx_internal_gmm_fused_create(int dim, int n_mix, int n_tiles, const void* A2_host, const uint32_t* mix_off,
                                             const uint32_t* k_mean, const double* c64, const float* means, const float* p1, const float* p2,
                                             void** rec_dev, size_t* rec_bytes) {
    const int    LD = amx::fused_ld(dim), REC = amx::fused_rec_bytes(dim);
    const size_t total = (size_t)n_tiles * REC;
    std::vector<char> rec(total, 0);
    const char*       A2 = (const char*)A2_host;
    for (int r = 0; r < n_tiles; ++r) {
        char* base = rec.data() + (size_t)r * REC;
        for (int row = 0; row < 256; ++row)
            for (int c = 0; c < 8; ++c)
                memcpy(base + row * 128 + ((c ^ ((row >> 1) & 7)) << 4), A2 + ((size_t)r * 256 + row) * 128 + c * 16, 16);
        float* pp = (float*)(base + amx::kFusedABytes);
        float* mu = (float*)(base + amx::kFusedAStage);
        for (int j = 0; j < 16; ++j) {
            const int m = r * 16 + j;
            int       nd = 0;
            if (m < n_mix) {
                nd = (int)(mix_off[m + 1] - mix_off[m]);
                for (int s = 0; s < nd; ++s) {
                    const uint32_t k   = mix_off[m] + s;
                    float*         dst = mu + (size_t)(j * 16 + s) * LD;
                    memcpy(dst, means + (size_t)k_mean[k] * dim, (size_t)dim * 4);
                    memcpy(dst + LD - 2, &c64[k], 8);
                }
            }
            pp[j]      = p1[m];  // p1 / p2 are padded to n_tiles * 16 entries
            pp[16 + j] = p2[m];
            ((int*)pp)[32 + j] = nd;
        }
    }
    void* d = nullptr;
    AMX_HIP(hipMalloc(&d, total));
    if (hipMemcpy(d, rec.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(d);
        amx::set_error("amx_gmm_create: upload of the fused tile records failed");
        return AMX_ERR_DEVICE;
    }
    *rec_dev   = d;
    *rec_bytes = total;
    return AMX_OK;
}

// how many mixture ranges a pass of Tpad frames is split into (one workgroup per CU and range); the partial arg-min arrays
// hold that many rows
extern "C" int amx_internal_gmm_fused_split(int n_cu, int Tpad, int n_tiles) {
    const int ntt = Tpad / 256;
    int       split = (std::max(n_cu, 8) + ntt - 1) / ntt;
    if (ntt * 4 >= 3 * std::max(n_cu, 8))
        split = 1;  // >= 3/4 of the CUs busy with whole-model workgroups: splitting would only add tile traffic
    return std::max(1, std::min(split, n_tiles));
}

extern "C" int amx_internal_gmm_fused_score(amx_ctx* ctx, int dim, const void* rec_dev, const float* isr_dev, const float* feats,
                                            const void* X, const float* nx, const float* q, int T, int Tpad, int n_mix, int n_tiles,
                                            int split, float* scores, uint32_t* best, float* pmin, unsigned* pidx, int part_ld,
                                            unsigned long long* survivors) {
    const int   ntt = Tpad / 256;
    const int   lds = 2 * amx::fused_rec_bytes(dim);
    const char* rec = (const char*)rec_dev;
    hipStream_t st  = ctx->stream;
    const int   abl = getenv("AMX_FUSED_ABL") ? atoi(getenv("AMX_FUSED_ABL")) : 0;  // ablation switches (profiling only; results are wrong)
#define AMX_FUSED(D)                                                                                                            \
    case D: {                                                                                                                   \
        if (best) {                                                                                                             \
            auto k = amx::gmm_fused_kernel<D, true>;                                                                            \
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                               \
            hipLaunchKernelGGL(k, dim3(ntt * split), dim3(512), lds, st, feats, (const _Float16*)X, nx, q, rec, isr_dev, scores, \
                               best, T, n_mix, n_tiles, split, pmin, pidx, part_ld, survivors, abl);                            \
        }                                                                                                                       \
        else {                                                                                                                  \
            auto k = amx::gmm_fused_kernel<D, false>;                                                                           \
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                               \
            hipLaunchKernelGGL(k, dim3(ntt * split), dim3(512), lds, st, feats, (const _Float16*)X, nx, q, rec, isr_dev, scores, \
                               best, T, n_mix, n_tiles, split, pmin, pidx, part_ld, survivors, abl);                            \
        }                                                                                                                       \
    } break;
    switch (dim) {
        AMX_FUSED(16)
        AMX_FUSED(24)
        AMX_FUSED(32)
        AMX_FUSED(33)
        AMX_FUSED(39)
        AMX_FUSED(40)
        default:
            amx::set_error("gmm fused scorer: no kernel for dimension %d", dim);
            return AMX_ERR_UNSUPPORTED;
    }
#undef AMX_FUSED
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}
