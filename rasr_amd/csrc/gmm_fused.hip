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
//     mean row (ds_read_b128, rows 16 B-slot staggered) fetched during the current distance -- and the ~2 % further survivors
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

// Instruction costs on MI355X (tools/valu_rates.hip, wall clock, all SIMDs busy): a SIMD saturates with TWO resident waves at
// ~1.2 ns per plain VOP2 f32 operation (sub / mul / add with VGPR operands), ~1.75 ns for VOP3 forms and for an SGPR operand,
// ~1.95 ns for f64 add / compare, and ~3.6 ns for v_pk_mul / v_pk_add / v_pk_fma_f32 -- a packed operation costs THREE plain ones,
// not two (a 1024-thread, 128-VGPR variant with four waves per SIMD was no faster: the VALU is the bound, not the occupancy).
// The reference's distance is therefore written with plain operations and this file is compiled with -fno-slp-vectorize (the
// SLP vectoriser would re-pack them).  Workgroup = 8 waves x 32 frames, two waves per SIMD, <= 256 VGPRs; per tile a wave runs the
// screen of its 32 frames, then the exact evaluation (the MFMA latencies of one wave hide behind the VALU work of the other).
#ifdef AMX_LAB
// lab builds: s_memtime stamps of workgroup 0, [wave < 16][tile < 64][phase]: 0 barrier passed (+ next tile's DMA issued), 1 screen
// done, 2 first survivors done, 3 further survivors done, 4 results issued; amx_lab_fused_stamps (tools/fused_timeline.py)
__device__ unsigned long long fused_stamps[16 * 64 * 6];
#define FUSED_STAMP(phase)                                                                                             \
    do {                                                                                                               \
        if (blockIdx.x == 0 && lane == 0 && (r - r_begin) < 64)                                                        \
            fused_stamps[(wave * 64 + (r - r_begin)) * 6 + (phase)] = __builtin_amdgcn_s_memtime();                    \
    } while (0)
#else
#define FUSED_STAMP(phase) \
    do {                   \
    } while (0)
#endif

// BEST: 0 scores only | 1 best density as u32 [T x n_mix] | 2 as ONE BYTE per (frame, mixture) (g_best is then a byte matrix with
// rows of n_mix bytes; 0xff where the u32 form writes 0xffffffff): a quarter of the index MEMORY, 10 MB instead of 40 per 1000 frames --
// the same time (what BEST costs this kernel is the vector work of tracking the index, 0.3 ms of 4.7, not the bytes)
// FMA: the reference's two builds (amx_gmm_model.tuning contract=off | fma, gmm_device.hpp sq_acc): false = `sum += df * df` as a
// product and a sum (-DMARCH=x86-64), true = as ONE v_fmac_f32, what the reference's default -march=native build executes on an FMA
// host -- three vector operations per dimension and survivor instead of four.  The screen does not change: its threshold bounds the
// distance of the f16 operands to the EXACT sum plus the (dim + 3) ulp of the unfused f32 evaluation, and the fused evaluation
// rounds half as often.
template<int DIM, int BEST, int NW, bool FMA = false>
__global__ __launch_bounds__(NW * 64) void gmm_fused_kernel(const float* __restrict__ g_feats, const _Float16* __restrict__ g_X,
                                                        const float* __restrict__ g_nx, const float* __restrict__ g_q,
                                                        const char* __restrict__ g_rec, const float* __restrict__ g_isr,
                                                        float* __restrict__ g_scores, uint32_t* __restrict__ g_best, int T, int Tpad,
                                                        int n_mix, int n_tiles, int r_split, float* __restrict__ g_part_min,
                                                        unsigned* __restrict__ g_part_idx, int part_ld,
                                                        unsigned long long* __restrict__ g_survivors, float na_all) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int LD = fused_ld(DIM), REC = fused_rec_bytes(DIM), NP = REC / 1024;
#ifdef FUSED_ISSUE_ALL
    constexpr bool ISSUE4 = false;
#else
    constexpr bool ISSUE4 = true;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_t = blockIdx.x / r_split, part = blockIdx.x % r_split;
    const int per = (n_tiles + r_split - 1) / r_split;
    const int r_begin = part * per, r_end = min(n_tiles, r_begin + per);
    if (r_begin >= r_end)
        return;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    // Which waves issue the record's LDS-DMA: with three waves per SIMD only the first of each SIMD (waves 0-3).  A piece holds its
    // wave until the address unit has taken it; the first wave of a SIMD wins the vector-issue arbitration, finishes every tile
    // first and waits a third of the period at the barrier (tools/fused_timeline.py) -- the issue time comes out of that slack
    // instead of out of every wave's start of the tile.
    constexpr int IW = (NW == 12 && ISSUE4) ? 4 : NW;
    auto load_tile = [&](int r, int buf) {
        if (wave >= IW)
            return;
        const char*    src = g_rec + (size_t)r * REC + lane * 16;
        const unsigned dst = lds_base + buf * REC;
        for (int p = wave; p < NP; p += IW)
            fused_dma16(src + p * 1024, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + p * 1024)));
    };
    load_tile(r_begin, 0);
#ifdef FUSED_PRIO  // lab: static priorities so that the three waves of a SIMD finish a tile together (the arbitration prefers the oldest)
    if (NW == 12) {
        if ((wave >> 2) == 2)
            __builtin_amdgcn_s_setprio(2);
        else if ((wave >> 2) == 1)
            __builtin_amdgcn_s_setprio(1);
    }
#endif
    const int  frow = lane & 31, fk = lane >> 5;
    const int  t    = tile_t * (NW * 32) + wave * 32 + frow;
    const bool live = t < T;
    const int  tt   = live ? t : T - 1;
    const int  tx   = min(t, Tpad - 1);  // packed operand rows exist up to Tpad
    const bool wave_live = tile_t * (NW * 32) + wave * 32 < T;  // wave-uniform
    float      x[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i)
        x[i] = g_feats[(size_t)tt * DIM + i];
    // K-steps of the screen: the operand rows are 64 f16 wide, but only DIM + 2 columns are used (a = -2 mu / sigma^2 and the constant as
    // c_hi + c_lo; pooled covariance) -- the columns behind them are zero in both operands, so their products are skipped: 3 matrix
    // instructions per block instead of 4 for dim 32-40, 2 for dim 16-24 (the masks, built from sums that only lose zero terms, stay the same)
    constexpr int KS = (DIM + 2 + 15) / 16;
    static_assert(KS >= 1 && KS <= 4, "screen operand rows hold 64 columns");
    fus_f16x8 bx[KS];
    float     nx, q;
    if (g_X) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            bx[ks] = *(const fus_f16x8*)(g_X + (size_t)tx * 64 + (ks * 2 + fk) * 8);
        nx = g_nx[tx];
        q  = g_q[tx];
    }
    else {
        // Round 6: the lane packs its frame's operand row itself (gmm_screen_pack_kernel's arithmetic on the row it already holds: a
        // launch and ~5 us less in front of a decoder-sized batch).  v = x / sigma, the f16 image of v, columns DIM and DIM + 1 = 1
        // against the constant's c_hi + c_lo, ||row|| and the residual norm behind the threshold's error bound; a frame behind the
        // last one packs zeros.  The two norms are summed in index order here and by a butterfly there: both are sums of 40
        // non-negative terms, a few ulp apart, under a threshold that is twice its bound.
        float n2 = 0.f, qq = 0.f, r2 = 0.f;
        bool  fits = true;
        _Float16 hv[KS * 16];
#pragma unroll
        for (int i = 0; i < KS * 16; ++i) {
            float v = 0.f;
            if (i < DIM)
                v = live ? x[i] * g_isr[i] : 0.f;
            fits &= fabsf(v) <= 65504.f;  // false for NaN as well
            const _Float16 h = (_Float16)v;
            const float    r = (float)h;
            n2 += r * r;
            r2 += (v - r) * (v - r);
            qq += v * v;
            hv[i] = (i == DIM || i == DIM + 1) ? (_Float16)(live ? 1.f : 0.f) : h;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                bx[ks][e] = fk ? hv[(ks * 2 + 1) * 8 + e] : hv[ks * 16 + e];
        nx = fits ? sqrtf(n2) : __builtin_inff();
        q  = 1.6e-5f * qq + na_all * (sqrtf(r2) * 1.00001f);
    }
    const bool  all = !(nx < __builtin_inff());  // operand row did not fit f16: keep every slot
    // stores of 16 bytes need aligned rows; otherwise (and on the model's last, partial tile) scalar guarded stores
    const bool wide_ok = (n_mix & (BEST == 2 ? 7 : 3)) == 0 && ((uintptr_t)g_scores & 15) == 0 &&
                         (!BEST || ((uintptr_t)g_best & (BEST == 2 ? 7 : 15)) == 0);
    const bool counted = wide_ok && wave_live && NW < 16;  // (the 16-wave instantiation spills: scratch traffic breaks the count)  // this wave issues exactly (4 | 3 | 2 for BEST = 1 | 2 | 0) stores per full tile
    // The compiler's wait-count pass does not see the inline-asm waits of the loop: without a wait it can see, it would put its own
    // vmcnt(0) in front of the first use of x[] inside the loop -- and there that waits for the next tile's DMA and the stores.
    __builtin_amdgcn_s_waitcnt(0);
    float    run_min = 3.402823466e+38f;  // best state of this lane's mixtures so far (ascending state, strict '<')
    unsigned run_idx = 0xffffffffu;
    unsigned n_surv  = 0;  // densities this lane evaluated exactly (bench: survivors per mixture)

    for (int r = r_begin; r < r_end; ++r) {
        const int buf = (r - r_begin) & 1;
        // my DMA pieces of tile r are older than the stores of tile r - 1: waiting until only those stores are outstanding
        // means the pieces have landed (gfx9 retires vector memory operations in issue order)
        if (wave >= IW)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // no DMA of its own in flight; its stores need no wait
        else if (r == r_begin || !counted)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (BEST == 1)
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (BEST == 2)
            asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everybody's pieces are there, and nobody reads the other stage any more
        if (r + 1 < r_end)
            load_tile(r + 1, buf ^ 1);
        FUSED_STAMP(0);
#ifdef FUSED_SLEEP  // lab: the three waves of a SIMD enter the screen one after the other instead of sharing the matrix pipe
        if (NW == 12) {
            if ((wave >> 2) == 1)
                __builtin_amdgcn_s_sleep(FUSED_SLEEP);
            else if ((wave >> 2) == 2)
                __builtin_amdgcn_s_sleep(2 * FUSED_SLEEP);
        }
#endif
        if (!wave_live)  // a wave behind the last frame only takes part in the DMA and the barriers
            continue;
        const char*  stage = lds + buf * REC;
        const float* s_p   = (const float*)(stage + kFusedABytes);  // p1[16], p2[16], (int) densities per mixture [16]

        // ---- screen: survivor masks of this lane's 8 mixtures (2 i + fk), 16 bit each
        unsigned M[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int  rr = i * 32 + frow;
            fus_f32x16 c;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                c[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const fus_f16x8 a = *(const fus_f16x8*)(stage + rr * 128 + (((ks * 2 + fk) ^ ((rr >> 1) & 7)) << 4));
                c                 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bx[ks], c, 0, 0, 0);
            }
            float mn = min3_first(c[0], c[1], c[2]);
            mn       = min3_raw(mn, c[3], c[4]);
            mn       = min3_raw(mn, c[5], c[6]);
            mn       = min3_raw(mn, c[7], c[8]);
            mn       = min3_raw(mn, c[9], c[10]);
            mn       = min3_raw(mn, c[11], c[12]);
            mn       = min3_raw(mn, c[13], c[14]);
            mn       = min3_raw(mn, c[15], c[15]);
            // tau as in gmm_screen_epilogue (gmm.hip): p1 = 2.05 ra + 1.3e-4 sqrtK, p2 = 1.3e-4 sqrtK na + 1.6e-5 cabs, q per frame
            const float p1 = s_p[i * 2 + fk], p2 = s_p[16 + i * 2 + fk];
            const int   nd = ((const int*)s_p)[32 + i * 2 + fk];
            const float thr = mn + fmaf(nx, p1, fmaf(fabsf(mn), 1.6e-5f, p2 + q)) + 1e-30f;
            unsigned    bits = 0;  // bit k = "slot k is above the threshold", filled from the top down
#pragma unroll
            for (int e = 15; e >= 0; --e)
                bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(thr - c[e]), 31);  // (bits << 1) | sign(thr - g)
            const unsigned valid = (1u << nd) - 1u;  // nd <= 16
            // frames behind the last one have an all-zero operand row (every slot ties): no survivors for them
            const unsigned m16 = live ? ((all ? 0xffffu : (~bits & 0xffffu)) & valid) : 0u;
            M[i >> 1] |= m16 << (16 * (i & 1));
        }
        n_surv += __popc(M[0]) + __popc(M[1]) + __popc(M[2]) + __popc(M[3]);
        FUSED_STAMP(1);

        // ---- exact evaluation.  First survivor of every mixture in lockstep (static register indices), the ~4 % further
        // survivors in a divergent loop behind it, in slot order.  The mean row comes in 8-float pieces (2 x ds_read_b128).
        const float* s_mu = (const float*)(stage + kFusedAStage);
        // the reference's distance (Mm/GaussDiagonalMaximumFeatureScorer.cc:143-180, SSE3 build): four strided partial sums of
        // unfused ((mu - x) / sigma)^2, (l0 + l1) + (l2 + l3), scalar tail -- plain f32 operations (see the cost table above)
        auto         distance = [&](int row, double& cc) -> float {
            const float*  src = s_mu + row * LD;
            float         l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
            constexpr int EFF = DIM & ~3;
#pragma unroll
            for (int i = 0; i < EFF; i += 4) {
                const float4 m  = *(const float4*)(src + i);
                const float  d0 = (m.x - x[i]) * g_isr[i], d1 = (m.y - x[i + 1]) * g_isr[i + 1];
                const float  d2 = (m.z - x[i + 2]) * g_isr[i + 2], d3 = (m.w - x[i + 3]) * g_isr[i + 3];
                l0              = sq_acc<FMA>(d0, l0);
                l1              = sq_acc<FMA>(d1, l1);
                l2              = sq_acc<FMA>(d2, l2);
                l3              = sq_acc<FMA>(d3, l3);
            }
            float result = 0.f;
            result       = result + ((l0 + l1) + (l2 + l3));
#pragma unroll
            for (int i = EFF; i < DIM; ++i) {
                const float df = (src[i] - x[i]) * g_isr[i];
                result         = sq_acc<FMA>(df, result);
            }
            cc = *(const double*)(src + LD - 2);
            return result;
        };
        float    best[8];
        unsigned bpack = 0, bvalid = 0;  // best density of mixture i in nibble i / bit i set once a density was taken
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned m16 = (M[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            const int      sl  = m16 ? __ffs((int)m16) - 1 : 0;
            double         cc;
            const float    dist = distance((i * 2 + fk) * 16 + sl, cc);
            const double   s    = cc + (double)dist;
            const bool     take = m16 != 0u && (double)FLT_MAX > s;  // reference: if (bestScore > score); bestScore is FLT_MAX so far
            best[i]             = take ? (float)s : FLT_MAX;
            bpack |= take ? ((unsigned)sl << (4 * i)) : 0u;
            bvalid |= take ? (1u << i) : 0u;
        }
        FUSED_STAMP(2);
        unsigned R[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {  // remove the first survivor of both halves
            const unsigned lo = M[w] & 0xffffu, hi = M[w] >> 16;
            R[w]              = (lo & (lo - 1u)) | ((hi & (hi - 1u)) << 16);
        }
        while (__any((R[0] | R[1] | R[2] | R[3]) != 0u)) {
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
                double    cc;
                const float dist = distance((i * 2 + fk) * 16 + sl, cc);
                float       b    = best[0];
#pragma unroll
                for (int j = 1; j < 8; ++j)
                    b = (i == j) ? best[j] : b;
                const double s    = cc + (double)dist;
                const bool   take = (double)b > s;
                const float  nb   = (float)s;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    best[j] = (take && i == j) ? nb : best[j];
                const unsigned sh = 4u * (unsigned)i;
                bpack             = take ? ((bpack & ~(15u << sh)) | ((unsigned)sl << sh)) : bpack;
                bvalid |= take ? (1u << i) : 0u;
            }
        }

        FUSED_STAMP(3);
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
        // lane halves exchange four values each: half 0 ends up with mixtures 0..7 of the tile, half 1 with 8..15, in order
        float    so[8];
        unsigned bo[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const fus_u32x2 rs = __builtin_amdgcn_permlane32_swap(__float_as_uint(sc[j]), __float_as_uint(sc[4 + j]), false, false);
            so[2 * j]          = __uint_as_float(rs.x);
            so[2 * j + 1]      = __uint_as_float(rs.y);
            if (BEST == 1) {
                const unsigned  bj = (bvalid >> j) & 1u ? (bpack >> (4 * j)) & 15u : 0xffffffffu;
                const unsigned  bk = (bvalid >> (4 + j)) & 1u ? (bpack >> (4 * (4 + j))) & 15u : 0xffffffffu;
                const fus_u32x2 rb = __builtin_amdgcn_permlane32_swap(bj, bk, false, false);
                bo[2 * j]          = rb.x;
                bo[2 * j + 1]      = rb.y;
            }
        }
        unsigned b8[2] = {0, 0};  // BEST == 2: the same eight indices as bytes, mixtures in order
        if (BEST == 2) {
            // 4-bit slots -> bytes (pairs 0..3 | 4..7), 0xff where no density was taken; ONE swap moves both words' halves
            auto nib2byte = [](unsigned n) {
                const unsigned x = (n | (n << 8)) & 0x00ff00ffu;
                return (x | (x << 4)) & 0x0f0f0f0fu;
            };
            auto bit2byte = [](unsigned v) { return ((v | (v << 7) | (v << 14) | (v << 21)) & 0x01010101u) * 0xffu; };
            const unsigned  wa = nib2byte(bpack & 0xffffu) | ~bit2byte(bvalid & 15u);
            const unsigned  wb = nib2byte(bpack >> 16) | ~bit2byte((bvalid >> 4) & 15u);
            const fus_u32x2 rb = __builtin_amdgcn_permlane32_swap(wa, wb, false, false);
            b8[0]              = __builtin_amdgcn_perm(rb.y, rb.x, 0x05010400u);  // x0 y0 x1 y1
            b8[1]              = __builtin_amdgcn_perm(rb.y, rb.x, 0x07030602u);  // x2 y2 x3 y3
        }
        const int mb = m0 + fk * 8;
        if (live) {
            float*    gs = g_scores + (size_t)t * n_mix + mb;
            uint32_t* gb = BEST == 1 ? g_best + (size_t)t * n_mix + mb : nullptr;
            unsigned char* gb8 = BEST == 2 ? (unsigned char*)g_best + (size_t)t * n_mix + mb : nullptr;
            if (wide_ok && m0 + 16 <= n_mix) {
                typedef float    nt_f4 __attribute__((ext_vector_type(4)));
                typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
                amx::nt_store(nt_f4{so[0], so[1], so[2], so[3]}, (nt_f4*)gs);
                amx::nt_store(nt_f4{so[4], so[5], so[6], so[7]}, (nt_f4*)(gs + 4));
                if (BEST == 1) {
                    amx::nt_store(nt_u4{bo[0], bo[1], bo[2], bo[3]}, (nt_u4*)gb);
                    amx::nt_store(nt_u4{bo[4], bo[5], bo[6], bo[7]}, (nt_u4*)(gb + 4));
                }
                if (BEST == 2) {
                    typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
                    // (plain stores, left to the L2 to merge into lines, measured the same: 4.59-4.63 against 4.58-4.62 ms)
                    amx::nt_store(nt_u2{b8[0], b8[1]}, (nt_u2*)gb8);
                }
            }
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (mb + e < n_mix) {
                        gs[e] = so[e];
                        if (BEST == 1)
                            gb[e] = bo[e];
                        if (BEST == 2)
                            gb8[e] = (unsigned char)(b8[e >> 2] >> (8 * (e & 3)));
                    }
            }
        }
        FUSED_STAMP(4);
    }
    if (g_survivors) {
        unsigned long long n = live ? n_surv : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            n += __shfl_xor(n, off, 64);
        if (lane == 0 && n)
            atomicAdd(g_survivors + ((blockIdx.x * NW + wave) & 255), n);  // 256 partial counters (summed by the host)
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


// ---------------------------------------------------------------------------------------------------------------------------------
// gmm_fused_spec_kernel: the same scorer with SPECIALISED waves (round-3 review item 4; amx_gmm_model.tuning fused_waves=13).
// tools/mx_probe.hip `coresident`: a wave that only issues matrix instructions and a wave that only runs the distance step overlap
// by 0.83 on one SIMD, while gmm_fused_kernel's waves all sit in the screen's MFMAs together and then all in the exact stage.  Here a
// workgroup is 4 SCREEN waves (one per SIMD) + NE = 8 or 12 EXACT waves (two or three per SIMD) over NE x 32 frames:
//   screen wave s   owns the frames of exact waves G s .. G s + G - 1 (G = 2 or 3 groups of 32 frames): their 32 MFMAs per tile and the mask
//                   epilogue -- exactly gmm_fused_kernel's screen, same masks -- and issues ALL of the workgroup's LDS-DMA;
//   exact wave e    owns 32 frames: reads its 4 mask words, then gmm_fused_kernel's exact stage, results and stores, unchanged.
// The screen runs ONE TILE AHEAD of the exact stage: in period r the screen waves work on tile r + 1 (its f16 rows arrived a period
// earlier) while the exact waves evaluate tile r from the masks of the period before.  LDS = gmm_fused_kernel's two records, now two
// rings of different phase: A ring [2][33 KB] (f16 rows + thresholds; A(q) in slot q & 1) and mean ring [2][mu stage]; the masks of
// tile r travel through the first 8 KB of A's slot r & 1 -- dead once every screen wave has finished tile r -- which takes three
// barriers per tile: A (screen of r + ... done, exact of r - 1 done) | screen waves write masks(r) | B | exact waves fetch them |
// C (slot free) | screen waves refill the slot by DMA and screen tile r + 1, exact waves evaluate tile r.
// MEASURED (tools/spec_test.py, 63 936 frames x 10 000 x 16): bit-identical to gmm_fused_kernel and SLOWER, 6.0 ms against 4.75 ms.
// The vector work per frame is the same in both kernels (75 against 73 instructions per frame and SIMD); two exact waves per SIMD do
// not keep the vector pipe as busy as three mixed waves do (their chains of dependent f32 / f64 operations and mean-row reads want a
// third wave to hide behind), and what the screen waves take off them was never on the critical path.  With twelve exact waves
// (NE = 12, sixteen waves of 128 registers) the kernel spills 448 bytes per lane: 11.7 ms.  Kept as an A/B variant (fused_waves=13)
// with its own parity test; gmm_fused_kernel stays the default.
template<int DIM, bool BEST, int NE>
__global__ __launch_bounds__((4 + NE) * 64) void gmm_fused_spec_kernel(const float* __restrict__ g_feats, const _Float16* __restrict__ g_X,
                                                             const float* __restrict__ g_nx, const float* __restrict__ g_q,
                                                             const char* __restrict__ g_rec, const float* __restrict__ g_isr,
                                                             float* __restrict__ g_scores, uint32_t* __restrict__ g_best, int T, int Tpad,
                                                             int n_mix, int n_tiles, int r_split, float* __restrict__ g_part_min,
                                                             unsigned* __restrict__ g_part_idx, int part_ld,
                                                             unsigned long long* __restrict__ g_survivors) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int LD = fused_ld(DIM), REC = fused_rec_bytes(DIM), A_ST = kFusedAStage, MU_ST = fused_mu_stage(DIM);
    constexpr int NS = 4, G = NE / NS, FPW = NE * 32;  // screen waves, frame groups per screen wave, frames per workgroup
    static_assert(NE % NS == 0, "every screen wave serves the same number of exact waves");
    static_assert(2 * REC == 2 * A_ST + 2 * MU_ST && A_ST % 1024 == 0 && MU_ST % 1024 == 0, "two rings out of two records");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_t = blockIdx.x / r_split, part = blockIdx.x % r_split;
    const int per = (n_tiles + r_split - 1) / r_split;
    const int r_begin = part * per, r_end = min(n_tiles, r_begin + per);
    if (r_begin >= r_end)
        return;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;
    auto a_slot  = [&](int q) { return lds + (q & 1) * A_ST; };
    auto mu_slot = [&](int q) { return lds + 2 * A_ST + (q & 1) * MU_ST; };
    const int frow = lane & 31, fk = lane >> 5;
    const int t_wg = tile_t * FPW;

    if (wave < NS) {
        // ============================================================ screen wave
        auto dma = [&](const char* src, unsigned dst, int n_pieces) {  // this wave's share of n_pieces KB
            for (int p = wave; p < n_pieces; p += NS)
                fused_dma16(src + p * 1024 + lane * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + p * 1024)));
        };
        auto load_a = [&](int r) {
            if (r < r_end)
                dma(g_rec + (size_t)r * REC, lds_base + (r & 1) * A_ST, A_ST / 1024);
        };
        auto load_mu = [&](int r) {
            if (r < r_end)
                dma(g_rec + (size_t)r * REC + A_ST, lds_base + 2 * A_ST + (r & 1) * MU_ST, MU_ST / 1024);
        };
        load_a(r_begin);
        load_mu(r_begin);
        load_a(r_begin + 1);
        // the two frame groups of this wave
        fus_f16x8 bx[G][4];
        float     nx[G], qq[G];
        bool      live[G], all[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int t  = t_wg + (G * wave + g) * 32 + frow;
            const int tx = min(t, Tpad - 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                bx[g][ks] = *(const fus_f16x8*)(g_X + (size_t)tx * 64 + (ks * 2 + fk) * 8);
            nx[g]   = g_nx[tx];
            qq[g]   = g_q[tx];
            live[g] = t < T;
            all[g]  = !(nx[g] < __builtin_inff());
        }
        unsigned M[G][4];
        auto     screen = [&](int r) {  // masks of tile r from A's slot r & 1: gmm_fused_kernel's screen, twice
            const char*  stage = a_slot(r);
            const float* s_p   = (const float*)(stage + kFusedABytes);
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    M[g][w] = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int  rr = i * 32 + frow;
                    fus_f32x16 c;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        c[e] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const fus_f16x8 a = *(const fus_f16x8*)(stage + rr * 128 + (((ks * 2 + fk) ^ ((rr >> 1) & 7)) << 4));
                        c                 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bx[g][ks], c, 0, 0, 0);
                    }
                    float mn = min3_first(c[0], c[1], c[2]);
                    mn       = min3_raw(mn, c[3], c[4]);
                    mn       = min3_raw(mn, c[5], c[6]);
                    mn       = min3_raw(mn, c[7], c[8]);
                    mn       = min3_raw(mn, c[9], c[10]);
                    mn       = min3_raw(mn, c[11], c[12]);
                    mn       = min3_raw(mn, c[13], c[14]);
                    mn       = min3_raw(mn, c[15], c[15]);
                    const float p1 = s_p[i * 2 + fk], p2 = s_p[16 + i * 2 + fk];
                    const int   nd = ((const int*)s_p)[32 + i * 2 + fk];
                    const float thr = mn + fmaf(nx[g], p1, fmaf(fabsf(mn), 1.6e-5f, p2 + qq[g])) + 1e-30f;
                    unsigned    bits = 0;
#pragma unroll
                    for (int e = 15; e >= 0; --e)
                        bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(thr - c[e]), 31);
                    const unsigned valid = (1u << nd) - 1u;
                    const unsigned m16   = live[g] ? ((all[g] ? 0xffffu : (~bits & 0xffffu)) & valid) : 0u;
                    M[g][i >> 1] |= m16 << (16 * (i & 1));
                }
            }
        };
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // P: A(r_begin), mean rows of r_begin, A(r_begin + 1) are there
        screen(r_begin);
        for (int r = r_begin; r < r_end; ++r) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // A: every screen wave is done with A(r); this period's DMA has landed; exact(r - 1) is done
#pragma unroll
            for (int g = 0; g < G; ++g)
                *(uint4*)(a_slot(r) + (((G * wave + g) * 32 + frow) * 2 + fk) * 16) = make_uint4(M[g][0], M[g][1], M[g][2], M[g][3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // B: masks(r) visible
            __builtin_amdgcn_s_barrier();  // C: the exact waves hold them: A's slot r & 1 is free
            load_a(r + 2);
            load_mu(r + 1);
            if (r + 1 < r_end)
                screen(r + 1);
        }
        return;
    }

    // ================================================================ exact wave
    const int  ew   = wave - NS;
    const int  t    = t_wg + ew * 32 + frow;
    const bool live = t < T;
    const int  tt   = live ? t : T - 1;
    const bool wave_live = t_wg + ew * 32 < T;
    float      x[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i)
        x[i] = g_feats[(size_t)tt * DIM + i];
    const bool wide_ok = (n_mix & 3) == 0 && ((uintptr_t)g_scores & 15) == 0 && (!BEST || ((uintptr_t)g_best & 15) == 0);
    __builtin_amdgcn_s_waitcnt(0);
    float    run_min = 3.402823466e+38f;
    unsigned run_idx = 0xffffffffu;
    unsigned n_surv  = 0;
    __builtin_amdgcn_s_barrier();  // P
    for (int r = r_begin; r < r_end; ++r) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // A
        __builtin_amdgcn_s_barrier();  // B: masks(r) are in A's slot r & 1
        const uint4 mw = *(const uint4*)(a_slot(r) + ((ew * 32 + frow) * 2 + fk) * 16);
        unsigned    M[4] = {mw.x, mw.y, mw.z, mw.w};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // C
        if (!wave_live)
            continue;
        n_surv += __popc(M[0]) + __popc(M[1]) + __popc(M[2]) + __popc(M[3]);
        const float* s_mu = (const float*)mu_slot(r);
        auto         distance = [&](int row, double& cc) -> float {
            const float*  src = s_mu + row * LD;
            float         l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
            constexpr int EFF = DIM & ~3;
#pragma unroll
            for (int i = 0; i < EFF; i += 4) {
                const float4 m  = *(const float4*)(src + i);
                const float  d0 = (m.x - x[i]) * g_isr[i], d1 = (m.y - x[i + 1]) * g_isr[i + 1];
                const float  d2 = (m.z - x[i + 2]) * g_isr[i + 2], d3 = (m.w - x[i + 3]) * g_isr[i + 3];
                l0              = l0 + d0 * d0;
                l1              = l1 + d1 * d1;
                l2              = l2 + d2 * d2;
                l3              = l3 + d3 * d3;
            }
            float result = 0.f;
            result       = result + ((l0 + l1) + (l2 + l3));
#pragma unroll
            for (int i = EFF; i < DIM; ++i) {
                const float df = (src[i] - x[i]) * g_isr[i];
                result         = result + df * df;
            }
            cc = *(const double*)(src + LD - 2);
            return result;
        };
        float    best[8];
        unsigned bpack = 0, bvalid = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned m16 = (M[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            const int      sl  = m16 ? __ffs((int)m16) - 1 : 0;
            double         cc;
            const float    dist = distance((i * 2 + fk) * 16 + sl, cc);
            const double   s    = cc + (double)dist;
            const bool     take = m16 != 0u && (double)FLT_MAX > s;
            best[i]             = take ? (float)s : FLT_MAX;
            bpack |= take ? ((unsigned)sl << (4 * i)) : 0u;
            bvalid |= take ? (1u << i) : 0u;
        }
        unsigned R[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned lo = M[w] & 0xffffu, hi = M[w] >> 16;
            R[w]              = (lo & (lo - 1u)) | ((hi & (hi - 1u)) << 16);
        }
        while (__any((R[0] | R[1] | R[2] | R[3]) != 0u)) {
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
                double    cc;
                const float dist = distance((i * 2 + fk) * 16 + sl, cc);
                float       b    = best[0];
#pragma unroll
                for (int j = 1; j < 8; ++j)
                    b = (i == j) ? best[j] : b;
                const double s    = cc + (double)dist;
                const bool   take = (double)b > s;
                const float  nb   = (float)s;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    best[j] = (take && i == j) ? nb : best[j];
                const unsigned sh = 4u * (unsigned)i;
                bpack             = take ? ((bpack & ~(15u << sh)) | ((unsigned)sl << sh)) : bpack;
                bvalid |= take ? (1u << i) : 0u;
            }
        }
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
        float    so[8];
        unsigned bo[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const fus_u32x2 rs = __builtin_amdgcn_permlane32_swap(__float_as_uint(sc[j]), __float_as_uint(sc[4 + j]), false, false);
            so[2 * j]          = __uint_as_float(rs.x);
            so[2 * j + 1]      = __uint_as_float(rs.y);
            if (BEST) {
                const unsigned  bj = (bvalid >> j) & 1u ? (bpack >> (4 * j)) & 15u : 0xffffffffu;
                const unsigned  bk = (bvalid >> (4 + j)) & 1u ? (bpack >> (4 * (4 + j))) & 15u : 0xffffffffu;
                const fus_u32x2 rb = __builtin_amdgcn_permlane32_swap(bj, bk, false, false);
                bo[2 * j]          = rb.x;
                bo[2 * j + 1]      = rb.y;
            }
        }
        const int mb = m0 + fk * 8;
        if (live) {
            float*    gs = g_scores + (size_t)t * n_mix + mb;
            uint32_t* gb = BEST ? g_best + (size_t)t * n_mix + mb : nullptr;
            if (wide_ok && m0 + 16 <= n_mix) {
                typedef float    nt_f4 __attribute__((ext_vector_type(4)));
                typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
                amx::nt_store(nt_f4{so[0], so[1], so[2], so[3]}, (nt_f4*)gs);
                amx::nt_store(nt_f4{so[4], so[5], so[6], so[7]}, (nt_f4*)(gs + 4));
                if (BEST) {
                    amx::nt_store(nt_u4{bo[0], bo[1], bo[2], bo[3]}, (nt_u4*)gb);
                    amx::nt_store(nt_u4{bo[4], bo[5], bo[6], bo[7]}, (nt_u4*)(gb + 4));
                }
            }
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (mb + e < n_mix) {
                        gs[e] = so[e];
                        if (BEST)
                            gb[e] = bo[e];
                    }
            }
        }
    }
    if (g_survivors) {
        unsigned long long n = live ? n_surv : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            n += __shfl_xor(n, off, 64);
        if (lane == 0 && n)
            atomicAdd(g_survivors + ((blockIdx.x * NE + ew) & 255), n);
    }
    if (g_part_min) {
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

// waves per workgroup: 12 (three per SIMD, 384 frames; the kernel uses 158 VGPRs) for long passes -- measured 5.1 ms against 5.9 ms
// with 8 waves and 5.0 ms with 16 (which spills 9 registers) per 63 936 frames -- and 8 (256 frames) for the decoder's small batches,
// where a 384-frame workgroup would idle a third of its waves.  amx_gmm_model.tuning fused_waves = 8 | 12 | 16 overrides (A/B runs).
static int fused_waves(int Tpad, int forced) {
    if (forced == 8 || forced == 12 || forced == 16 || forced == 13)  // 13: the specialised kernel (4 screen + 8 exact waves)
        return forced;
    return Tpad >= 4096 ? 12 : 8;
}
static int fused_frames(int Tpad, int forced) {  // frames per workgroup
    const int w = fused_waves(Tpad, forced);
    return w == 13 ? 256 : w * 32;
}

// how many mixture ranges a pass of Tpad frames is split into (workgroup = its frames x one range; one workgroup per CU): the
// smallest split that gives every CU a workgroup, unless the frame tiles alone already fill 3/4 of them.  The partial arg-min
// arrays hold that many rows.
static int fused_split_raw(int n_cu, int Tpad, int n_tiles, int forced) {
    const int fpw = fused_frames(Tpad, forced);
    const int ntt = (Tpad + fpw - 1) / fpw, cus = std::max(n_cu, 8);
    if (fused_waves(Tpad, forced) != 8) {  // frame tiles of 384: pick the split whose workgroup count is closest below a whole number of rounds
        int best = 1;
        double best_eff = 0;
        for (int sp = 1; sp <= std::min(n_tiles, 8); ++sp) {
            const int    wgs = ntt * sp;
            const double eff = (double)wgs / (double)(((wgs + cus - 1) / cus) * cus);
            if (eff > best_eff + 0.02) {
                best_eff = eff;
                best     = sp;
            }
        }
        return best;
    }
    int split = (cus + ntt - 1) / ntt;
    if (ntt * 4 >= 3 * cus)
        split = 1;
    else if (ntt * split > cus && split > 1 && ntt * (split - 1) * 8 >= cus * 7)
        --split;  // one workgroup per CU and a few idle CUs beat a second, nearly empty round
    return std::max(1, std::min(split, n_tiles));
}

// waves per workgroup the pass will run with (the byte-sized best-density output exists for 8 and 12)
extern "C" int amx_internal_gmm_fused_waves(int Tpad, int forced_waves) {
    return fused_waves(Tpad, forced_waves);
}

extern "C" int amx_internal_gmm_fused_split(int n_cu, int Tpad, int n_tiles, int forced_waves) {
    // every range holds ceil(n_tiles / split) tiles: report the number of NON-EMPTY ranges, the partial arg-min arrays have
    // exactly that many rows (an empty range would leave its row unwritten)
    const int split = fused_split_raw(n_cu, Tpad, n_tiles, forced_waves);
    const int per   = (n_tiles + split - 1) / split;
    return (n_tiles + per - 1) / per;
}

extern "C" int amx_internal_gmm_fused_score(amx_ctx* ctx, int dim, const void* rec_dev, const float* isr_dev, const float* feats,
                                            const void* X, const float* nx, const float* q, int T, int Tpad, int n_mix, int n_tiles,
                                            int split, float* scores, uint32_t* best, float* pmin, unsigned* pidx, int part_ld,
                                            unsigned long long* survivors, int forced_waves, int best_bytes, int contract_fma, float na_all) {
    const int   nw = fused_waves(Tpad, forced_waves), fpw = fused_frames(Tpad, forced_waves), ntt = (Tpad + fpw - 1) / fpw;
    const int   lds = 2 * amx::fused_rec_bytes(dim);
    const char* rec = (const char*)rec_dev;
    hipStream_t st  = ctx->stream;
#define AMX_FUSED_LAUNCH(D, B, W)                                                                                               \
    {                                                                                                                           \
        auto k = contract_fma ? amx::gmm_fused_kernel<D, B, W, true> : amx::gmm_fused_kernel<D, B, W, false>;                   \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                   \
        hipLaunchKernelGGL(k, dim3(ntt * split), dim3(W * 64), lds, st, feats, (const _Float16*)X, nx, q, rec, isr_dev, scores,  \
                           best, T, Tpad, n_mix, n_tiles, split, pmin, pidx, part_ld, survivors, na_all);                  \
    }
#define AMX_FUSED_SPEC(D, B, E)                                                                                                 \
    {                                                                                                                           \
        auto k = amx::gmm_fused_spec_kernel<D, B, E>;                                                                           \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                   \
        hipLaunchKernelGGL(k, dim3(ntt * split), dim3((4 + E) * 64), lds, st, feats, (const _Float16*)X, nx, q, rec, isr_dev, scores,    \
                           best, T, Tpad, n_mix, n_tiles, split, pmin, pidx, part_ld, survivors);                          \
    }
#define AMX_FUSED(D)                                                                                                            \
    case D: {                                                                                                                   \
        if (nw == 13 && best) AMX_FUSED_SPEC(D, true, 8)                                                                        \
        else if (nw == 13) AMX_FUSED_SPEC(D, false, 8)                                                                          \
        else if (best && nw == 16) AMX_FUSED_LAUNCH(D, 1, 16)                                                                   \
        else if (best && nw == 12 && best_bytes == 1) AMX_FUSED_LAUNCH(D, 2, 12)                                                \
        else if (best && best_bytes == 1) AMX_FUSED_LAUNCH(D, 2, 8)                                                             \
        else if (best && nw == 12) AMX_FUSED_LAUNCH(D, 1, 12)                                                                   \
        else if (best) AMX_FUSED_LAUNCH(D, 1, 8)                                                                                \
        else if (nw == 16) AMX_FUSED_LAUNCH(D, 0, 16)                                                                           \
        else if (nw == 12) AMX_FUSED_LAUNCH(D, 0, 12)                                                                           \
        else AMX_FUSED_LAUNCH(D, 0, 8)                                                                                          \
    } break;
    if (nw == 13 && !X) {
        amx::set_error("gmm fused scorer: the specialised-wave kernel (fused_waves=13) reads packed operand rows (fused_pack=0)");
        return AMX_ERR_UNSUPPORTED;
    }
    if (nw == 13 && contract_fma) {
        amx::set_error("gmm fused scorer: the specialised-wave kernel (fused_waves=13) exists for contract=off only");
        return AMX_ERR_UNSUPPORTED;
    }
    if (best && best_bytes == 1 && nw != 8 && nw != 12) {
        amx::set_error("gmm fused scorer: byte-sized best densities exist for the 8- and 12-wave kernels only (fused_waves=%d)", nw);
        return AMX_ERR_UNSUPPORTED;
    }
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
#undef AMX_FUSED_SPEC
#undef AMX_FUSED_LAUNCH
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

#ifdef AMX_LAB
extern "C" int amx_lab_fused_stamps(unsigned long long* out /* [16 * 64 * 6] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(amx::fused_stamps), sizeof(amx::fused_stamps)) == hipSuccess ? AMX_OK : AMX_ERR_DEVICE;
}
#endif
