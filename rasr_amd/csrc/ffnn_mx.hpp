// ffnn_mx.hpp -- AMX_PREC_F16MX: the round-4 arithmetic of the NN leg (included by ffnn.hip, after gemm_epilogue).
//
// An f32 product x w of the reference's sgemm (Nn/LinearLayer.cc:298-324 -> Math/Blas.hh:402-420) is taken as
//     h(x) h(w)                        h = f16(v), round to nearest              v_mfma_f32_32x32x16_f16        (1 unit of matrix time)
//   + q(x) r(w) + r(x) q(w)            q = fp6(h(v)), r = fp6(v - h(v))          v_mfma_scale_f32_32x32x64_f8f6f4, e2m3 x e2m3
// with OCP-MX block scales (one e8m0 exponent per 32 k and row: 2^(E-2) for q, E the exponent of the block maximum, and 2^-11 of
// that for r -- |v - f16(v)| <= 2^-11 2^E, so r never saturates and needs no maximum of its own).  WEIGHT rows n and n ^ 32 share
// their exponent (the maximum of the two blocks): a lane of the GEMM serves both (two 32-row blocks of its wave tile), the conversion
// instruction converts 32 values under ONE scale, so one instruction yields q for both rows.  Frames keep an exponent of their own
// (paired like the rows a frame's scores would depend on which frame shares its batch, and scoring a segment in pieces would no
// longer give the bits of scoring it whole); there the lane's two frames trade k-halves with the partner lane instead
// (v_permlane32_swap: a lane then holds all 32 k of ONE frame, see products()).  Three conversions per K-tile and wave of 128 x 64
// instead of six; they cost: tools/cvt_rate.hip measures 28 ns per conversion and SIMD and NO overlap with the matrix instructions
// (16 + 8 of them: 675 ns; with six conversions: 966 ns).  Because both cross terms carry
// the same total scale 2^(Ew-2) 2^(Ex-13), ONE 64-deep scaled product does both for 16 k:
//     A = [ q(w) (16 k) | r(w) (the same 16 k) ]   scale 2^(Ew-2)
//     B = [ r(x)        | q(x)                 ]   scale 2^(Ex-13)
// and the 16 k of a lane are the 16 k its two f16 fragments of the K-tile hold anyway (lanes 0-31: chunks 0 and 2 of a row's 32 k,
// lanes 32-63: chunks 1 and 3), so q is NOT stored: it is converted from the fragment registers (v_cvt_scalef32_pk32_fp6_f16, one
// instruction per fragment pair).  Memory holds, per value, 2 bytes of f16 and 6 bits of r (+ one scale byte per 16 values):
// 3 B, against 4 B for split bf16 -- and 1.5 units of matrix time per product instead of 3 (fp6 x fp6 runs at four times the f16
// rate: 32 cycles per 32x32x64).  The dropped r r term is <= 2^-22 |x w|; the cross terms carry a relative error of ~2^-4 on a
// 2^-11 term.  tools/emulate_split_f16_f8.py evaluates the scheme on all 10.24 M scores of BASELINE config 4
// (profiles/r04/emulation_f16_f8.json): worst |d| = 0.029 of the 1e-4 |ref| + 1e-4 bar, no arg-min change; the fp4 form (0.11 of
// the bar, built first) failed the bar on a one-output network whose scores are not carried by a prior (tests/test_ffnn_f16mx_gpu.py).
// f16 range: an activation or feature beyond +-65504 raises the handle's overflow flag and every later call fails (use bf16x3).
//
// Memory layout (weights and activations alike; rows = output units or frames, padded to 256; K padded to 32):
//   one 24 KB block per (256 rows, K-tile of 32 k), blocks ordered [row block][K-tile]; a block is the LDS image of the tile:
//     H  @0      [256 rows][64 B]        f16, the row's four 16-byte chunks XOR-swizzled by (row >> 2) & 3 (conflict-free ds_read_b128)
//     R  @16384  [2 halves][256 rows][16 B]   per (row, half): 16 e2m3 fields (12 B) = r of the half's 16 k in fragment order, then
//                                        one dword whose low byte is the scale the row brings to the scaled product
//                                        (weights: E - 2, activations: E - 13)
//   so every LDS-DMA piece (one wave-wide 16-byte global_load_lds = 1 KB) reads 1 KB of CONTIGUOUS memory -- eight whole cache
//   lines -- instead of 16 row pieces of 64 B at an 8 KB stride, and a line is used up by one K-tile.
//   Order of the 32 k inside a tile (free, as long as both operands agree): natural index u = 8 g + 4 hh + e is what an MFMA
//   accumulator lane (hh = lane >> 5) holds in registers (g, e) of a 32 x 32 result block; the f16 plane keeps it at position
//   pos16(u): lane (frame, hh) of the hidden-layer epilogue owns f16 chunks hh and 2 + hh and the whole R record of half hh, and
//   writes them straight from its registers (no exchange, no LDS round trip).
#pragma once

namespace amx {
namespace mx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef float    f32v16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef int      v8i __attribute__((ext_vector_type(8)));

constexpr int TK    = 32;     // k per K-tile
constexpr int R_OFF = 16384;  // residual records
constexpr int BLK   = 24576;  // bytes per (256 rows, K-tile)

__host__ __device__ constexpr int pos16(int u) {  // position of natural index u in the f16 plane (8 per 16-byte chunk)
    return 8 * (2 * (u >> 4) + ((u >> 2) & 1)) + 4 * ((u >> 3) & 1) + (u & 3);
}
__host__ __device__ inline int h_off(int r, int c) {  // 16-byte chunk c of row r (0..255) of a block's f16 plane
    return r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
}
__host__ __device__ inline int r_off(int r, int half) {  // residual record of (row, half)
    return R_OFF + half * 4096 + r * 16;
}

// biased exponent E of the block maximum, at least 14 (E - 13 stays the exponent field of a normal float)
__host__ __device__ inline int block_exponent(float max_abs) {
    unsigned u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(max_abs);
#else
    memcpy(&u, &max_abs, 4);
#endif
    const int e = (int)((u >> 23) & 0xffu);
    return e < 14 ? 14 : e;
}

// ---- host side (weights, once): e2m3 code of v / 2^(sbyte - 127), round to nearest even, saturating -- what the device
// conversions do (tools/build/fp6_probe, cvt_probe)
static inline unsigned fp6_code_host(float v, int sbyte) {
    const double a = std::fabs((double)v) * std::ldexp(1.0, 127 - sbyte);
    unsigned     code;
    if (a >= 7.5)
        code = 31;
    else {
        const double step = a >= 4.0 ? 0.5 : a >= 2.0 ? 0.25 : 0.125;  // subnormals and the binade [1, 2) share the step 1/8
        const double qv   = std::nearbyint(a / step) * step;            // default rounding mode: to nearest even
        if (qv >= 4.0)  // the rounded value may have crossed into the next binade
            code = 24 + (unsigned)((qv - 4.0) / 0.5);
        else if (qv >= 2.0)
            code = 16 + (unsigned)((qv - 2.0) / 0.25);
        else
            code = (unsigned)(qv / 0.125);
    }
    return code | (std::signbit(v) ? 32u : 0u);
}

// packs rows [n_rows x K] (f32, row stride ld) into blocks [Rpad / 256][KT]; rows / columns beyond the matrix are zero
static inline void pack_weights_host(const float* W, int n_rows, int K, int ld, int Rpad, int KT, std::vector<unsigned char>& out) {
    out.assign((size_t)(Rpad / 256) * KT * BLK, 0);
    auto block_max = [&](int n, int kt) {
        float m = 0.f;
        if (n < n_rows)
            for (int u = 0; u < 32; ++u) {
                const int k = kt * 32 + u;
                if (k < K)
                    m = std::fmax(m, std::fabs(W[(size_t)n * ld + k]));
            }
        return m;
    };
    for (int n = 0; n < n_rows; ++n) {
        const int rb = n >> 8, r = n & 255;
        for (int kt = 0; kt < KT; ++kt) {
            unsigned char* blk = out.data() + ((size_t)rb * KT + kt) * BLK;
            float          v[32], lo[32];
            _Float16       hi[32];
            for (int u = 0; u < 32; ++u) {
                const int k = kt * 32 + u;
                v[u]        = k < K ? W[(size_t)n * ld + k] : 0.f;
                hi[u]       = (_Float16)v[u];
                lo[u]       = v[u] - (float)hi[u];
            }
            const int ec = block_exponent(std::fmax(block_max(n, kt), block_max(n ^ 32, kt)));  // one exponent for rows n and n ^ 32
            unsigned  rec[2][4] = {{0, 0, 0, (unsigned)(ec - 2)}, {0, 0, 0, (unsigned)(ec - 2)}};
            for (int u = 0; u < 32; ++u) {
                const int p = pos16(u), chunk = p >> 3, half = chunk & 1, f = 8 * (chunk >> 1) + (p & 7);  // field f of the half's record
                memcpy(blk + h_off(r, chunk) + (p & 7) * 2, &hi[u], 2);
                const unsigned code = fp6_code_host(lo[u], ec - 13);
                const int      bit = 6 * f, d = bit >> 5, sh = bit & 31;
                rec[half][d] |= code << sh;
                if (sh > 26)
                    rec[half][d + 1] |= code >> (32 - sh);
            }
            memcpy(blk + r_off(r, 0), rec[0], 16);
            memcpy(blk + r_off(r, 1), rec[1], 16);
        }
    }
}

// 16 values of one accumulator lane (natural indices u = 8 g + 4 hh + e at a[4 g + e]: the half's fragment order) -> its two f16
// chunks and its residual record (scale byte: sbyte)
struct LanePack {
    uint4 h0, h1, rec;
};

__device__ __forceinline__ unsigned pk_f16(float a, float b) {
    f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ LanePack lane_pack(const float (&a)[16], int ec, int sbyte) {
    LanePack o;
    unsigned h[8];
    f32v16   le, lod;  // residuals of the even / odd fields (the conversion interleaves its two inputs)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        h[p]           = pk_f16(a[2 * p], a[2 * p + 1]);
        const f16x2 hv = __builtin_bit_cast(f16x2, h[p]);
        le[p]          = a[2 * p] - (float)hv[0];
        lod[p]         = a[2 * p + 1] - (float)hv[1];
        le[8 + p]      = 0.f;
        lod[8 + p]     = 0.f;
    }
    o.h0 = make_uint4(h[0], h[1], h[2], h[3]);
    o.h1 = make_uint4(h[4], h[5], h[6], h[7]);
    // The conversion WRITES ITS FIRST RESULT DWORD BEFORE IT HAS READ ALL 32 INPUTS, and this compiler does not know: through the
    // builtin it gave the result the registers of input elements 4-9 (v[6:11] out of v[2:17]) and field 8 came out as garbage
    // (first seen in the tanh instantiation only, where the allocator happened to overlap them).  Inline assembly with an
    // early-clobber result forbids the overlap; the trailing s_nop covers the wait states the compiler's hazard recogniser would
    // put between a vector write and a matrix-instruction read of the result (it does not look into an asm statement).
    u32x6       r;
    const float scale = __uint_as_float((unsigned)(ec - 13) << 23);
    asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3\n\ts_nop 2" : "=&v"(r) : "v"(le), "v"(lod), "v"(scale));
    o.rec = make_uint4(r[0], r[1], r[2], (unsigned)sbyte);
    return o;
}

// one lane's share of row r of block `blk` (hh = which half of the 32 k it holds)
__device__ __forceinline__ void lane_store(char* blk, int r, int hh, const LanePack& p) {
    *(uint4*)(blk + h_off(r, hh))     = p.h0;
    *(uint4*)(blk + h_off(r, 2 + hh)) = p.h1;
    *(uint4*)(blk + r_off(r, hh))     = p.rec;
}

// f32 frames [T x K] (row stride ldx) -> blocks [Tpad / 256][KT].  Two lanes per (frame, K-tile): lane half hh holds the natural
// indices 8 g + 4 hh + e, exactly like an accumulator lane of the GEMM epilogue, so both go through lane_pack / lane_store.
__global__ __launch_bounds__(256) void pack_input_mx(const float* __restrict__ x, int ldx, int T, int K, char* __restrict__ out, int KT, int Tpad,
                                                    unsigned* __restrict__ overflow) {
    const long long n = (long long)Tpad * KT * 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int       hh = (int)(i & 1);
        const long long j  = i >> 1;
        const int       kt = (int)(j % KT), t = (int)(j / KT);
        float           a[16], m = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int   k = kt * 32 + 8 * g + 4 * hh + e;
                const float v = (t < T && k < K) ? x[(size_t)t * ldx + k] : 0.f;
                a[4 * g + e]  = v;
                m             = (v != v) ? __builtin_inff() : fmaxf(m, fabsf(v));  // a NaN counts as out of range (fmaxf would drop it)
            }
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        if (!(m < 65520.f))  // beyond the f16 range, or not finite (fmaxf drops a NaN: see nan below)
            *overflow = 1u;
        const int ec = block_exponent(m);
        lane_store(out + ((size_t)(t >> 8) * KT + kt) * BLK, t & 255, hh, lane_pack(a, ec, ec - 13));
    }
}

// in place: v -> act(-v) (the last hidden layer run through the score epilogue, -(W x + b): exact)
template<int ACT>
__global__ __launch_bounds__(256) void neg_act_kernel(float* __restrict__ x, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        x[i] = activate<ACT>(-x[i]);
}

// ---------------------------------------------------------------------------------------------- tile configurations
template<int BN_, int BT_, int WN_, int WT_, int STAGES_, int PF_ = 0, int IW_ = 0, int U_ = 1, int LW_ = 0, int PIPE_ = 0>
struct MxCfg {
    static constexpr int BN = BN_, BT = BT_, WN = WN_, WT = WT_, STAGES = STAGES_;
    // PIPE = 1 (round 5): PING-PONG halves.  Waves w and w + 4 share a SIMD; a K-tile's period is cut into two halves by a SECOND
    // barrier, and in each half one wave of every SIMD issues its products while its partner does everything that is not matrix work
    // -- the fragment reads of its next K-tile, its share of the refill's LDS-DMA pieces -- then the roles swap:
    //     half 1 of K-tile kt    waves 0-3: products(kt)                      waves 4-7: refill share, reads(kt)
    //     half 2                 waves 0-3: refill share, reads(kt + 1)       waves 4-7: products(kt)
    // Round 4's ablations of this kernel added up exactly (fragment reads + barriers 0.76 ms, conversions 0.44, DMA issue 0.25, matrix
    // work 0.8 of the output layer's 2.2 ms): with all eight waves in phase nothing overlapped, and without the second barrier (the
    // SKEW variant below) the two groups drift until both sit in the same kind of segment again.  Here a SIMD's matrix pipe is handed
    // from one wave to the other at every barrier.  No second register image (a wave reads its next K-tile into the registers whose
    // products were issued a half period earlier), the same three-stage ring with the same two K-tiles in flight (a stage is read by
    // the first group in half 2 of period kt - 1 and by the second in half 1 of period kt, and is refilled behind that), the same
    // matrix instructions in the same order per accumulator: bit-identical scores.
    static constexpr int PIPE = PIPE_;
    // PIPE = 2: ONE WAVE PER SIMD (four waves, each a 128 x 128 wave tile: 256 accumulator registers in the AGPR half of a 512-register
    // file) that pipelines itself -- see the K loop.  Against the eight-wave tile: a third fewer fragment bytes out of LDS per K-tile
    // (4 x (128 + 128) rows instead of 8 x (64 + 128)), four conversions per SIMD and K-tile instead of six (a conversion takes the
    // matrix pipe), one barrier per K-tile among four waves instead of two among eight.
    // PIPE = 3 (round 6): READ-AHEAD for the one-tile-per-CU configuration with loader waves (U K-tiles per barrier).  Round 6's ablations
    // of a 2048 x 2048 layer at batch 1024 (profiles/r06/hidden_layer_probe.log): 33 us as is, 33 us with NO operand DMA behind the
    // prologue, 30 us with no matrix instructions, 30 us with neither -- the period of an iteration is the computing wave's own chain
    // barrier -> fragment reads -> LDS latency -> conversions -> products -> barrier, one wave per SIMD and nothing to overlap it with.
    // Here a computing wave reads the fragments of K-tile k + 1 into a SECOND register image while it issues the products of K-tile k
    // -- one or two LDS reads behind every matrix instruction (a burst of 18 reads holds an in-order wave as long as waiting for them
    // did: measured, no gain).  The ring keeps its schedule with one more K-tile awaited per barrier (see the K loop).  Per accumulator
    // the same matrix instructions in the same order: bit-identical scores.
    static_assert(PIPE_ == 0 || (PIPE_ == 1 && U_ == 1 && LW_ == 0 && PF_ == 0 && IW_ == 0 && STAGES_ == 3 && WN_ * WT_ == 8) ||
                      (PIPE_ == 2 && U_ == 1 && LW_ == 0 && PF_ == 0 && IW_ == 0 && STAGES_ == 3 && WN_ * WT_ == 4 && BN_ / WN_ == 128 && BT_ / WT_ == 128) ||
                      (PIPE_ == 3 && U_ == 2 && LW_ > 0 && STAGES_ >= 2 * U_ + 1),
                  "ping-pong K loop: eight waves, every one of them issuing, plain three-stage ring; self-pipelined: four waves of 128 x 128; "
                  "read-ahead: loader waves, two K-tiles per barrier, a ring of at least three iterations");
    // PF > 0 (256 x 256 tiles only): waves 0-5 touch the 384 cache lines of the K-tile PF steps ahead of the one whose LDS-DMA they
    // have just issued (one dword per 128-byte line, 64 lines per wave-instruction) -- a software prefetch from the Infinity Cache /
    // HBM into L2.  The LDS ring holds two K-tiles in flight (~2 periods of 1.7 us); a K-tile whose lines miss L2 (23 % of the
    // requests, pmc/pmc_l2a.txt) arrives later than that and every wave ends its period waiting (tools/mx_timeline.py: 500-1600 of
    // 3500 cycles per K-tile in s_waitcnt vmcnt).
    static constexpr int PF = PF_;
    // U: K-tiles per barrier.  Small-batch tiles (one 128 x 64 tile per CU, 6 + 3 matrix instructions per wave and K-tile) spend a
    // K-tile's period on the barrier and the LDS round trip, not on arithmetic: with U = 2 a wave reads two K-tiles into two
    // register images behind ONE barrier and issues both sets of products (still in ascending K: bit-identical results).
    static constexpr int U = U_;
    static_assert(U_ >= 1 && U_ <= 4 && STAGES_ >= 2 * U_, "ring: U K-tiles being read + at least U in flight");
    static_assert(PF_ == 0 || (BN_ == 256 && BT_ == 256 && WN_ * WT_ == 8), "the prefetch walks whole 256-row blocks with six waves");
    // LW > 0 (hidden layers only): LW extra LOADER waves that issue every LDS-DMA piece and nothing else; the NW compute waves never
    // touch the address unit.  A piece holds its issuing wave ~85 cycles (tools/mx_timeline.py small: 850 of the 2450 cycles of a
    // two-K-tile period of a 128 x 64 tile went into a compute wave's ten pieces, one wave per SIMD and nothing to overlap with).
    static constexpr int NW = WN * WT, LW = LW_, THREADS = (NW + LW_) * 64, PLANES = 1;
    static_assert(LW_ == 0 || (IW_ == 0 && PF_ == 0), "loader waves replace the issuing-wave variants");
    // K-loop variants of gemm_mx_kernel, both measured on the output layer (profiles/r04/gemm_mx_ablation3.log: 2.17-2.31 ms in
    // all four combinations, i.e. no gain) and left off: SKEW = the two waves of a SIMD half a K-tile apart, SPREAD = the LDS-DMA
    // pieces issued between the matrix instructions instead of as a burst behind the barrier.  Lab builds flip them (DBG 256 / 512).
    static constexpr bool SKEW = false, SPREAD = false;
    static constexpr int MI = BN / WN / 32, MJ = BT / WT / 32;
    static constexpr int A_R = BN * 64, A_BYTES = BN * 96;  // [H: 64 B/row][R: 2 halves x 16 B/row]
    static constexpr int B_R = BT * 64, B_BYTES = BT * 96;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES, LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int A_PIECES = BN / 16 + BN / 32, B_PIECES = BT / 16 + BT / 32;  // 1 KB pieces of H and R
    static constexpr int TOTAL = A_PIECES + B_PIECES;
    // IW: waves that issue the LDS-DMA (default all).  IW = NW / 2: the first wave of every SIMD issues all pieces, its partner none
    // -- the partner goes from its fragment reads straight to its products while the issuing wave sits in the address unit's queue
    static constexpr int IW  = LW_ > 0 ? LW_ : IW_ > 0 ? IW_ : NW;
    static constexpr int PPW = (TOTAL + IW - 1) / IW;  // pieces per issuing wave, at most
    static constexpr int NHI = TOTAL % IW;             // waves 0 .. NHI-1 issue PPW pieces, the other issuing waves PPW - 1 (NHI == 0: all PPW)
    // PF with IW = NW / 2 (the FREE form): the waves that issue no LDS-DMA issue the prefetch loads and never wait on vmcnt inside the
    // K-loop -- the loads float freely, nothing queues behind them.  PF with IW = NW: waves 0-5 carry one load per refill in their
    // counted queue (measured 6 % slower: a load that misses L2 holds back the count of the next K-tile's pieces).
    static constexpr bool PF_FREE = PF_ > 0 && IW_ > 0;
    static_assert(PF_ == 0 || IW_ == 0 || IW_ * 2 == WN_ * WT_, "free prefetch: half of the waves issue the DMA");
    static_assert(BN % 64 == 0 && BT % 64 == 0 && 256 % BN == 0 && 256 % BT == 0, "tiles are whole record pieces of a 256-row block");
    static_assert(STAGES >= 2 && STAGES <= 8 && (STAGES - 2) * (PPW + (PF_ > 0 ? 1 : 0)) + 1 < 64, "vmcnt immediate");
};

// LDS-DMA piece p of a K-tile: source offset inside the operand's block (ha / hb: which part of the 256 rows the tile covers),
// destination offset inside the stage, operand
template<class C>
__device__ __forceinline__ void piece_offsets(int p, int ha, int hb, int& src, int& dst, bool& is_b) {
    is_b = p >= C::A_PIECES;
    if (!is_b) {
        if (p < C::BN / 16) {
            src = (ha * (C::BN / 16) + p) * 1024;
            dst = p * 1024;
        }
        else {
            const int j = p - C::BN / 16, half = j / (C::BN / 64), gq = j % (C::BN / 64);
            src = R_OFF + half * 4096 + (ha * (C::BN / 64) + gq) * 1024;
            dst = C::A_R + half * (C::BN * 16) + gq * 1024;
        }
    }
    else {
        const int q = p - C::A_PIECES;
        if (q < C::BT / 16) {
            src = (hb * (C::BT / 16) + q) * 1024;
            dst = C::A_BYTES + q * 1024;
        }
        else {
            const int j = q - C::BT / 16, half = j / (C::BT / 64), gq = j % (C::BT / 64);
            src = R_OFF + half * 4096 + (hb * (C::BT / 64) + gq) * 1024;
            dst = C::A_BYTES + C::B_R + half * (C::BT * 16) + gq * 1024;
        }
    }
}

#ifdef AMX_LAB
// lab builds (DBG 2048): s_memtime stamps of workgroup 0's first tile, [wave][K-tile < 48][phase]: 0 barrier passed, 1 refill issued,
// 2 fragments in registers, 3 products issued; read back with amx_lab_mx_stamps (tools/mx_timeline.py)
__device__ unsigned long long mx_stamps[8 * 48 * 4];
// tile-level stamps of workgroup 0, wave 0: for its first four tiles {s_memtime, s_memrealtime} at tile start, K-loop end, tile end
__device__ unsigned long long mx_tile_stamps[4 * 3 * 2];
#endif

// ablation builds: keep a register value alive without using it (plain __device__ functions: the host pass does not look at their asm;
// as a template the substitution failed on the host -- silently, and the kernel's host stub was never emitted)
__device__ __forceinline__ void keep_alive(const v8i& v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void keep_alive(const f16x8& v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void keep_alive(unsigned v) { asm volatile("" ::"v"(v)); }

template<int N>
__device__ __forceinline__ void mx_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// s_waitcnt vmcnt(ahead * PER + EXTRA) for ahead = 0 .. MAXA (0: everything; the immediate must be a constant: one branch per value)
template<int PER, int EXTRA, int MAXA>
__device__ __forceinline__ void mx_wait_ahead(int ahead) {
    if constexpr (MAXA <= 0)
        mx_wait<0>();
    else {
        if (ahead >= MAXA)
            mx_wait<MAXA * PER + EXTRA>();
        else
            mx_wait_ahead<PER, EXTRA, MAXA - 1>(ahead);
    }
}

// q of a lane's 16 k of TWO rows (n, n ^ 32: they share the block exponent) from their four f16 fragments in ONE conversion:
// dwords 0-2 = the first row's fields, 3-5 = the second row's
__device__ __forceinline__ u32x6 q_fields_pair(f16x8 c0, f16x8 c1, f16x8 d0, f16x8 d1, unsigned scale_byte) {
    const f16x16 v = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const f16x16 u = __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const f16x32 w = __builtin_shufflevector(v, u, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
    u32x6        q;  // early-clobber result: see lane_pack
    const float  scale = __uint_as_float(scale_byte << 23);
    asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2\n\ts_nop 2" : "=&v"(q) : "v"(w), "v"(scale));
    return q;
}

// q of a lane's 16 k from its two f16 fragments (a wave tile with ONE 32-row block on that side).  FIRST: the fields go to dwords 0-2
// of the operand (the A side), else to dwords 3-5 (the B side); the other half of the conversion's input is left undefined.
template<bool FIRST>
__device__ __forceinline__ u32x6 q_fields(f16x8 c0, f16x8 c1, unsigned scale_byte) {
    const f16x16 v = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    // both halves hold the 16 values (round 6): an UNDEFINED half may be given any registers -- it got the destinations of fragment reads
    // still in flight (the read-ahead loop), and the compiler's s_waitcnt lgkmcnt in front of the conversion waited for them.  The
    // fields of the other half are not used either way.
    const f16x32 w = __builtin_shufflevector(v, v, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    u32x6       q;  // early-clobber result: see lane_pack
    const float scale = __uint_as_float(scale_byte << 23);
    asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2\n\ts_nop 2" : "=&v"(q) : "v"(w), "v"(scale));
    return q;
}

// D[n][t] = sum_k W[n][k] X[t][k] in the arithmetic above.  W, X: blocks (see the head of the file; xkts = K-tiles per row block
// of X); hidden layers write the next layer's blocks (ktn K-tiles per row block), the output layer f32 scores [T x n_valid] =
// -(D + bias) with the arg-min partials of the fused statistics.  Persistent workgroups, XCD-aware tile order, STAGES-deep LDS
// ring with ONE barrier per K-tile and counted vmcnt (never a drain inside the loop).
// DBG (lab builds only, -DAMX_LAB + AMX_MX_DBG): 8 no matrix instructions, 16 no operand DMA after the prologue, 32 no scaled product
// (conversions and MX MFMAs skipped), 64 every workgroup streams one of 8 tiles (all operands L2 hits), 128 no fp6 conversions, 256 wave skew, 512 LDS-DMA pieces spread between the products, 1024 rotated K walk per tile, 2048 s_memtime stamps of workgroup 0 (mx_stamps), 4096 fragment reads awaited in front of the refill burst
// ksplit (round 5, opt-in: amx_ffnn_model.tuning ksplit=4; small batches only): SPLIT-K ACROSS WORKGROUPS, in two launches.  A decoder's
// buffer fill (256 frames) gives a 2048 x 2048 layer 64 tiles of 128 x 64 -- 64 of the 256 CUs stream 1.2 MB of operands each, at the
// ~47 GB/s ONE CU's LDS ring pulls from the Infinity Cache (ring bytes in flight / memory latency: Little's law -- not arithmetic, and not
// the barrier chain: splitting K among the waves of one workgroup was built first and was slower, it pulls the same bytes through the
// same ring).  ksplit > 1: launch 1, `ksplit` workgroups per tile, each walks its share of K and parks its accumulators in a workspace
// (no epilogue); ksplit < -1: launch 2, one workgroup per tile adds the |ksplit| partial sums in group order, ((P0 + P1) + P2) + P3, and
// runs the epilogue.  (One launch with an arrival counter and the last workgroup reducing was built too: bit-identical, and 3 x slower
// than not splitting -- the eight L2s are not coherent with one another, so the device-scope fences around the counter write back and
// invalidate a whole L2 per workgroup.  A kernel boundary does that once.)  The sum over k is associated differently from the default
// (one accumulator, ascending k): scores differ from the default's by f32 rounding -- both meet the 1e-4 bar against the
// f64-accumulating reference sum -- and are bit-identical among all passes that run split (every pass of at most 256 frames of a handle
// created with ksplit=4): a decoder runs one buffer size for life.
template<class C, int ACT, bool LAST, int DBG = 0>
__global__ __launch_bounds__(C::THREADS) void gemm_mx_kernel(const char* __restrict__ W, const char* __restrict__ X, const float* __restrict__ bias,
                                                            void* __restrict__ out, int KT, int xkts, int ktn, int ldo, int n_valid, int t_valid,
                                                            int n_tiles_n, int n_tiles_total, int GT, int GN, float* __restrict__ part_min,
                                                            unsigned* __restrict__ part_idx, int part_ld, unsigned* __restrict__ overflow, int stagger,
                                                            int ksplit, float* __restrict__ ks_ws) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [STAGES][A: H | R][B: H | R] ... [bias]
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn   = (wave % C::NW) / C::WT, wt = wave % C::WT;  // (a loader wave's tile coordinates are never used)
    constexpr int WNR = C::BN / C::WN, WTT = C::BT / C::WT;
    const bool    loader  = C::LW > 0 && wave >= C::NW;             // issues the LDS-DMA, takes part in the barriers, computes nothing
    const int     iwave   = C::LW > 0 ? wave - C::NW : wave;        // index among the issuing waves (negative: not one of them)
    const bool    issuer  = iwave >= 0 && iwave < C::IW;
    const bool    hi_wave = C::NHI == 0 || iwave < C::NHI;          // issues PPW pieces per K-tile

    // Staggered start (tuning "stagger" > 0, in 10 ns ticks per XCD; off by default).  All workgroups run tiles of the same length, so
    // they reach their epilogues together: 64 MB of scores per round of tiles leave in one burst (tools/mx_timeline.py: 9-10 us of
    // a 125 us tile).  XCD x starts x * stagger late, so the bursts of the eight XCDs would follow one another -- measured: no
    // change at 1 / 3 / 6 us per XCD (profiles/r04/stagger.log); the non-temporal stores already drain behind the next tile's K loop.
    if (stagger > 0) {
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        const unsigned long long ticks   = (unsigned long long)(blockIdx.x & 7) * (unsigned)stagger;
        while (__builtin_amdgcn_s_memrealtime() - t_start < ticks)
            __builtin_amdgcn_s_sleep(8);
    }
    // split-K exists for the one-tile-per-CU configurations only (the host passes ksplit = 1 to every other one): compiled out elsewhere,
    // where the reduction's code cost the 256 x 256 tiles registers they do not have
    constexpr bool KSPLIT = C::BN == 128 && C::BT == 64 && C::U == 2;
    const int  ks     = KSPLIT ? (ksplit > 1 ? ksplit : 1) : 1;   // launch 1: K groups per tile
    const bool reduce = KSPLIT && ksplit < -1;                    // launch 2: partial sums -> epilogue
    const int n_units = n_tiles_total * ks;  // ks > 1: unit = (tile, K group), the K groups of a tile on neighbouring workgroups
    for (int ui = blockIdx.x; ui < n_units; ui += gridDim.x) {
        const int vi = KSPLIT ? ui / ks : ui, kgroup = KSPLIT ? ui - vi * ks : 0;
        int tile_t, tile_n;
        {  // the XCD-aware order of gemm_bf16_kernel
            const int nwg = n_tiles_total;
            const int q = nwg >> 3, r = nwg & 7, xcd = vi & 7, k = vi >> 3;
            const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
            const int n_tiles_t = nwg / n_tiles_n;
            if (GT > 0 && GN > 0) {
                const int band = v / (GT * n_tiles_n), w = v - band * (GT * n_tiles_n);
                const int bt0  = band * GT, bh = min(GT, n_tiles_t - bt0);
                const int blk  = w / (bh * GN), u = w - blk * (bh * GN);
                const int bn0  = blk * GN, bw = min(GN, n_tiles_n - bn0);
                tile_t         = bt0 + u / bw;
                tile_n         = bn0 + u % bw;
            }
            else {
                tile_t = v / n_tiles_n;
                tile_n = v - tile_t * n_tiles_n;
            }
        }
        if constexpr ((DBG & 64) != 0) {
            tile_t = blockIdx.x & 7;
            tile_n = 0;
        }
        auto tile_stamp = [&](int phase) {
#ifdef AMX_LAB
            if constexpr ((DBG & 2048) != 0) {
                const int nth = (vi - (int)blockIdx.x) / (int)gridDim.x;
                if (blockIdx.x == 0 && nth < 4 && tid == 0 && (!LAST || n_tiles_total > 1000)) {
                    mx_tile_stamps[(nth * 3 + phase) * 2]     = __builtin_amdgcn_s_memtime();
                    mx_tile_stamps[(nth * 3 + phase) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
                }
            }
#endif
        };
        tile_stamp(0);
        const int   n0 = tile_n * C::BN, t0 = tile_t * C::BT;
        const char* wblk = W + (size_t)(n0 >> 8) * KT * BLK;
        const char* xblk = X + (size_t)(t0 >> 8) * xkts * BLK;  // xkts >= KT: the producer padded its outputs to 256
        const int   ha = (n0 & 255) / C::BN, hb = (t0 & 255) / C::BT;
        // split-K: this workgroup walks K-tiles [kt_lo, kt_lo + KT) of the KT_all (whole multiples of U per group); the K loop below
        // is the usual one on shifted operand pointers
        const int KT_all = KT;
        if (reduce)
            KT = 0;   // no K loop: the accumulators come from the workspace
        if (KSPLIT && ks > 1) {
            const int per = ((KT_all + ks - 1) / ks + C::U - 1) / C::U * C::U, kt_lo = min(kgroup * per, KT_all);
            KT   = min(per, KT_all - kt_lo);
            wblk += (size_t)kt_lo * BLK;
            xblk += (size_t)kt_lo * BLK;
        }

        // ---- this wave's pieces of a K-tile (wave-uniform: scalar registers)
        int  p_src[C::PPW], p_dst[C::PPW];
        bool p_b[C::PPW];
#pragma unroll
        for (int q = 0; q < C::PPW; ++q) {
            const int p = q * C::IW + iwave;
            if (p < C::TOTAL && issuer)
                piece_offsets<C>(p, ha, hb, p_src[q], p_dst[q], p_b[q]);
            else {
                p_src[q] = p_dst[q] = 0;
                p_b[q]              = false;
            }
        }
        // 1 KB per wave-instruction straight into LDS (LDS address = M0 + lane * 16; global address = scalar base + lane * 16).  Inline
        // assembly on purpose: for an LDS-DMA it knows about, the compiler's wait-count pass puts s_waitcnt vmcnt(0) in front of the
        // first LDS read it cannot prove disjoint -- here a record read of the CURRENT stage, i.e. every K-tile drained the two K-tiles
        // in flight (found in the ISA; the ablation table of profiles/r04/gemm_mx_ablation.log was taken with that drain in place).
        // The ordering is explicit: counted vmcnt + s_barrier at the top of a K-tile.  Nothing else in the kernel uses M0.
        const unsigned voff = (unsigned)lane * 16u;
        const unsigned lds_base = (unsigned)(uintptr_t)lds;
        auto piece = [&](int q, int slot, int kt) {  // q: compile-time constant after unrolling
            if ((q == C::PPW - 1 && !hi_wave) || !issuer)
                return;
            int ktm = kt;
            if constexpr ((DBG & 1024) != 0) {  // ablation: every tile starts its walk over K somewhere else (other L2 channels)
                ktm = kt + (tile_n * 5 + tile_t * 3) % KT;
                ktm = ktm >= KT ? ktm - KT : ktm;
            }
            const char*    src = (p_b[q] ? xblk : wblk) + (size_t)ktm * BLK + p_src[q];
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + slot * C::STAGE_BYTES + p_dst[q]));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory");
        };
        float    pf_reg  = 0.f;  // destination of the prefetch loads: stays allocated for the whole K-loop (a load may land any time)
        const bool pf_wave = C::PF > 0 && (C::PF_FREE ? wave >= C::IW : wave < 6);
        auto prefetch = [&](int kt) {  // K-tile kt of the operand stream: 8 KB chunks (64 lines) 0-2 of W's block, 3-5 of X's
            if constexpr (C::PF > 0) {
                if (pf_wave) {
                    const int ktc = min(kt, KT - 1);
                    if constexpr (C::PF_FREE) {  // four waves share the six chunks: waves IW, IW + 1 take two
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const int ch = (wave - C::IW) + 4 * c;
                            if (ch < 6) {
                                const char* p = (ch < 3 ? wblk : xblk) + (size_t)ktc * BLK + (ch % 3) * 8192;
                                asm volatile("global_load_dword %0, %1, %2" : "+v"(pf_reg) : "v"((unsigned)lane * 128u), "s"(p) : "memory");
                            }
                        }
                    }
                    else {
                        const char* p = (wave < 3 ? wblk : xblk) + (size_t)ktc * BLK + (wave % 3) * 8192;
                        asm volatile("global_load_dword %0, %1, %2" : "+v"(pf_reg) : "v"((unsigned)lane * 128u), "s"(p) : "memory");
                    }
                }
            }
        };
        auto stage = [&](int slot, int kt) {
#pragma unroll
            for (int q = 0; q < C::PPW; ++q)
                piece(q, slot, kt);
            prefetch(kt + C::PF);   // (piece() returns at once for a wave that issues no DMA: the free form's prefetching waves land here too)
        };

        f32x16 acc[C::MI][C::MJ];
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::MJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][r] = 0.f;
        float* s_bias = (float*)(lds + gemm_scratch_bytes<C, LAST>());
        for (int i = tid; i < C::BN; i += C::THREADS)
            s_bias[i] = bias[n0 + i];

#pragma unroll
        for (int s = 0; s < ((C::PIPE == 1 || C::PIPE == 2) ? C::STAGES : C::STAGES - C::U); ++s)
            if (s < KT)
                stage(s, s);
        const int frow = lane & 31, fk = lane >> 5;
        const int a_row = wn * WNR + frow, b_row = wt * WTT + frow;
        // Register image of one K-tile: the f16 fragments of both k-slabs and the residual records.
        struct Frag {
            f16x8 a[2][C::MI], b[2][C::MJ];
            uint4 ra[C::MI], rb[C::MJ];
        };
        Frag fr[C::U];  // (read-ahead, PIPE 3: the two images swap roles every K-tile)
        auto reads = [&](int kt, Frag& F) {
            auto& a  = F.a;
            auto& b  = F.b;
            auto& ra = F.ra;
            auto& rb = F.rb;
            const char* ab = lds + (kt % C::STAGES) * C::STAGE_BYTES;
            const char* bb = ab + C::A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < C::MI; ++i)
                    a[ks][i] = *(const f16x8*)(ab + h_off(a_row + 32 * i, 2 * ks + fk));
#pragma unroll
                for (int j = 0; j < C::MJ; ++j)
                    b[ks][j] = *(const f16x8*)(bb + h_off(b_row + 32 * j, 2 * ks + fk));
            }
#pragma unroll
            for (int i = 0; i < C::MI; ++i)
                ra[i] = __builtin_bit_cast(uint4, *(const f16x8*)(ab + C::A_R + fk * (C::BN * 16) + (a_row + 32 * i) * 16));  // typed like the fragment reads: a uint4 read made the compiler drain vmcnt in front of it
#pragma unroll
            for (int j = 0; j < C::MJ; ++j)
                rb[j] = __builtin_bit_cast(uint4, *(const f16x8*)(bb + C::B_R + fk * (C::BT * 16) + (b_row + 32 * j) * 16));
        };
        // SPREAD variant: the LDS-DMA pieces of the K-tile that refills the freed stage are issued BETWEEN the matrix instructions,
        // GAP apart (a piece holds the issuing wave for 100-200 cycles while the address unit takes its 1 KB, and a wave issues in
        // order).  Measured: no gain over the burst behind the barrier (MxCfg).
        constexpr int N_MFMA = 3 * C::MI * C::MJ, GAP = N_MFMA / C::PPW > 0 ? N_MFMA / C::PPW : 1;  // matrix instructions per piece
        int           dma_kt = -1;  // K-tile whose pieces the next products() issues (-1: none)
        // READ-AHEAD (PIPE 3): the fragment reads of K-tile side_kt go out BETWEEN the matrix instructions of the K-tile in front of
        // them, into the other register image (side_F; nullptr: none).  A wave issues in order and the LDS takes its time over 18 wide
        // reads (450 cycles for the four computing waves' 72 KB, tools/mx_timeline.py small): issued as one burst they held the
        // matrix instructions back exactly as long as waiting for them had -- the read-ahead alone changed nothing
        // (profiles/r06/hidden_layer_probe.log).  One or two reads behind each matrix instruction ride in its shadow.
        // (the image is passed by reference, the switch as a flag: a captured pointer to it sent all four images to scratch)
        constexpr int N_SIDE = 3 * (C::MI + C::MJ);  // reads of one K-tile: a[0][.] b[0][.] a[1][.] b[1][.] ra[.] rb[.]
        auto side_read = [&](Frag& F, int side_kt, int p) {  // p: compile-time constant after unrolling
            const char* ab = lds + (side_kt % C::STAGES) * C::STAGE_BYTES;
            const char* bb = ab + C::A_BYTES;
            if (p < C::MI)
                F.a[0][p] = *(const f16x8*)(ab + h_off(a_row + 32 * p, fk));
            else if (p < C::MI + C::MJ)
                F.b[0][p - C::MI] = *(const f16x8*)(bb + h_off(b_row + 32 * (p - C::MI), fk));
            else if (p < 2 * C::MI + C::MJ)
                F.a[1][p - C::MI - C::MJ] = *(const f16x8*)(ab + h_off(a_row + 32 * (p - C::MI - C::MJ), 2 + fk));
            else if (p < 2 * (C::MI + C::MJ))
                F.b[1][p - 2 * C::MI - C::MJ] = *(const f16x8*)(bb + h_off(b_row + 32 * (p - 2 * C::MI - C::MJ), 2 + fk));
            else if (p < 3 * C::MI + 2 * C::MJ)
                F.ra[p - 2 * (C::MI + C::MJ)] = __builtin_bit_cast(uint4, *(const f16x8*)(ab + C::A_R + fk * (C::BN * 16) + (a_row + 32 * (p - 2 * (C::MI + C::MJ))) * 16));
            else
                F.rb[p - 3 * C::MI - 2 * C::MJ] = __builtin_bit_cast(uint4, *(const f16x8*)(bb + C::B_R + fk * (C::BT * 16) + (b_row + 32 * (p - 3 * C::MI - 2 * C::MJ)) * 16));
        };
        auto products_side = [&](Frag& F, Frag& SIDE, int side_kt, bool side_on) {
            auto& a  = F.a;
            auto& b  = F.b;
            auto& ra = F.ra;
            auto& rb = F.rb;
            int n_issued = 0;  // compile-time after unrolling
            auto after_mfma = [&]() {
                if constexpr (C::PIPE == 3) {
                    if (side_on && !(DBG & 4)) {   // (DBG 4, lab: no fragment reads behind the first K-tile's -- stale registers, timing only)
                        const int lo = n_issued * N_SIDE / N_MFMA, hi = (n_issued + 1) * N_SIDE / N_MFMA;
#pragma unroll
                        for (int p = 0; p < N_SIDE; ++p)
                            if (p >= lo && p < hi)
                                side_read(SIDE, side_kt, p);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (C::SPREAD != ((DBG & 512) != 0)) {
                    if (n_issued % GAP == GAP / 2 && n_issued / GAP < C::PPW) {
                        if (dma_kt >= 0)
                            piece(n_issued / GAP, dma_kt % C::STAGES, dma_kt);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                ++n_issued;
            };
            // the same order in every configuration: h.h of k-slab 0, of k-slab 1, then the scaled cross product
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < C::MI; ++i)
#pragma unroll
                    for (int j = 0; j < C::MJ; ++j) {
                        if constexpr ((DBG & 8) != 0)
                        {
                            keep_alive(a[ks][i]);
                            keep_alive(b[ks][j]);
                        }
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
                        after_mfma();
                    }
            // (A scheduling barrier here, keeping the conversions behind all sixteen f16 products -- the scheduler hoists the first of
            // them behind the FIRST product, where it waits for the records, the last reads of the K-tile -- measured 2.20 vs 2.13 ms:
            // slower.  The hoisted conversions run beside the matrix instructions.)
            if constexpr ((DBG & 32) != 0) {
#pragma unroll
                for (int i = 0; i < C::MI; ++i)
                    keep_alive(ra[i].x), keep_alive(ra[i].y), keep_alive(ra[i].z), keep_alive(ra[i].w);
#pragma unroll
                for (int j = 0; j < C::MJ; ++j)
                    keep_alive(rb[j].x), keep_alive(rb[j].y), keep_alive(rb[j].z), keep_alive(rb[j].w);
                for (int m = 0; m < C::MI * C::MJ; ++m)
                    after_mfma();
                if (dma_kt >= 0)
                    prefetch(dma_kt + C::PF);
                dma_kt = -1;
                return;
            }
            // q(w) = fp6(h(w) / 2^(Ew - 2)), q(x) = fp6(h(x) / 2^(Ex - 2)) (the x record holds Ex - 13).  Rows 32 i and 32 (i + 1) of
            // the wave tile (i even) are a pair n, n ^ 32 with one exponent: one conversion for both
            v8i av[C::MI], bv[C::MJ];
            if constexpr (C::MI % 2 == 0) {
#pragma unroll
                for (int i = 0; i < C::MI; i += 2) {
                    u32x6 q;
                    if constexpr ((DBG & 128) != 0)
                        q = u32x6{ra[i].y, ra[i].z, ra[i].x, ra[i + 1].y, ra[i + 1].z, ra[i + 1].x};  // ablation: no conversion
                    else
                        q = q_fields_pair(a[0][i], a[1][i], a[0][i + 1], a[1][i + 1], ra[i].w);
                    av[i]     = v8i{(int)q[0], (int)q[1], (int)q[2], (int)ra[i].x, (int)ra[i].y, (int)ra[i].z, 0, 0};
                    av[i + 1] = v8i{(int)q[3], (int)q[4], (int)q[5], (int)ra[i + 1].x, (int)ra[i + 1].y, (int)ra[i + 1].z, 0, 0};
                }
            }
            else {
#pragma unroll
                for (int i = 0; i < C::MI; ++i) {
                    u32x6 q;
                    if constexpr ((DBG & 128) != 0)
                        q = u32x6{ra[i].y, ra[i].z, ra[i].x, 0, 0, 0};
                    else
                        q = q_fields<true>(a[0][i], a[1][i], ra[i].w);
                    av[i] = v8i{(int)q[0], (int)q[1], (int)q[2], (int)ra[i].x, (int)ra[i].y, (int)ra[i].z, 0, 0};
                }
            }
            // Frames keep an exponent each, and a lane holds HALF of a frame's 32 k (chunks fk and 2 + fk) for its two frames t and
            // t + 32.  v_permlane32_swap trades the upper lanes' first frame for the lower lanes' second one: the lower lane then holds
            // all four chunks of frame t, the upper lane those of frame t + 32 -- 32 values under one exponent, ONE conversion for the
            // pair of blocks -- and three more swaps hand the fields of the other half back.  Same inputs, same scales, same fields as
            // the two half-filled conversions: bit-identical.
            if constexpr (C::MJ % 2 == 0 && (DBG & 128) == 0) {
#pragma unroll
                for (int j = 0; j < C::MJ; j += 2) {
                    u32x4 x0[2], x1[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        x0[ks] = __builtin_bit_cast(u32x4, b[ks][j]);
                        x1[ks] = __builtin_bit_cast(u32x4, b[ks][j + 1]);
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const u32x2 sw = __builtin_amdgcn_permlane32_swap(x0[ks][d], x1[ks][d], false, false);
                            x0[ks][d]      = sw[0];  // lanes 0-31: frame t chunk 2 ks;     lanes 32-63: frame t + 32 chunk 2 ks
                            x1[ks][d]      = sw[1];  // lanes 0-31: frame t chunk 2 ks + 1; lanes 32-63: frame t + 32 chunk 2 ks + 1
                        }
                    }
                    const unsigned sc = (fk ? rb[j + 1].w : rb[j].w) + 11u;
                    u32x6          q  = q_fields_pair(__builtin_bit_cast(f16x8, x0[0]), __builtin_bit_cast(f16x8, x0[1]), __builtin_bit_cast(f16x8, x1[0]),
                                                      __builtin_bit_cast(f16x8, x1[1]), sc);  // dwords 0-2: chunks 0, 2; 3-5: chunks 1, 3 of the lane's frame
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const u32x2 sw = __builtin_amdgcn_permlane32_swap(q[d], q[3 + d], false, false);
                        q[d]           = sw[0];  // the lane's own half (chunks fk, 2 + fk) of frame t
                        q[3 + d]       = sw[1];  // ... of frame t + 32
                    }
                    bv[j]     = v8i{(int)rb[j].x, (int)rb[j].y, (int)rb[j].z, (int)q[0], (int)q[1], (int)q[2], 0, 0};
                    bv[j + 1] = v8i{(int)rb[j + 1].x, (int)rb[j + 1].y, (int)rb[j + 1].z, (int)q[3], (int)q[4], (int)q[5], 0, 0};
                }
            }
            else {
#pragma unroll
                for (int j = 0; j < C::MJ; ++j) {
                    u32x6 q;
                    if constexpr ((DBG & 128) != 0)
                        q = u32x6{0, 0, 0, rb[j].y, rb[j].z, rb[j].x};
                    else
                        q = q_fields<false>(b[0][j], b[1][j], rb[j].w + 11u);
                    bv[j] = v8i{(int)rb[j].x, (int)rb[j].y, (int)rb[j].z, (int)q[3], (int)q[4], (int)q[5], 0, 0};
                }
            }
#pragma unroll
            for (int i = 0; i < C::MI; ++i)
#pragma unroll
                for (int j = 0; j < C::MJ; ++j) {
                    if constexpr ((DBG & 8) != 0)
                    {
                        keep_alive(av[i]);
                        keep_alive(bv[j]);
                    }
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av[i], bv[j], acc[i][j], 2, 2, 0, (int)ra[i].w, 0, (int)rb[j].w);
                    after_mfma();
                }
            if (dma_kt >= 0)
                prefetch(dma_kt + C::PF);  // SPREAD variant: the refill's prefetch follows its last piece, as in stage()
            dma_kt = -1;
        };
        auto products = [&](Frag& F) { products_side(F, F, 0, false); };
        // PIPE 3, the order that lets ONE wave per SIMD overlap with itself (64 x 32 wave tile: two f16 products per k-slab, one pair
        // conversion for the rows, one for the frames, two scaled products).  Round 6's ablations (profiles/r06/hidden_layer_probe.log):
        // with neither fragment reads nor matrix instructions nor operand DMA the K loop still took 15 of its 26 us -- a conversion's
        // RESULT was awaited right behind its issue (the v_mov that assemble the scaled operands), behind the f16 products.  Here the
        // two conversions are ISSUED FIRST (their inputs -- every fragment and both scale bytes of the K-tile -- arrived under the
        // previous K-tile's products), the four f16 products and the next K-tile's reads follow while they run, and the operands are
        // assembled in front of the scaled products.  Per accumulator: f16 k-slab 0, f16 k-slab 1, scaled -- as everywhere.
        auto products_ahead = [&](Frag& F, Frag& SIDE, int side_kt, bool side_on) {
            static_assert(C::PIPE != 3 || (C::MI == 2 && C::MJ == 1), "read-ahead K loop: 64 x 32 wave tiles");
            if constexpr (C::MI == 2 && C::MJ == 1) {
                u32x6 qa, qb;
                if constexpr ((DBG & 128) != 0) {
                    qa = u32x6{F.ra[0].y, F.ra[0].z, F.ra[0].x, F.ra[1].y, F.ra[1].z, F.ra[1].x};
                    qb = u32x6{0, 0, 0, F.rb[0].y, F.rb[0].z, F.rb[0].x};
                }
                else {
                    qa = q_fields_pair(F.a[0][0], F.a[1][0], F.a[0][1], F.a[1][1], F.ra[0].w);
                    qb = q_fields<false>(F.b[0][0], F.b[1][0], F.rb[0].w + 11u);
                }
                __builtin_amdgcn_sched_barrier(0);
                // (Assembling the operands of the scaled products BETWEEN the f16 products -- empty asm statements that make each
                // tuple exist there -- was measured too: 27.4 -> 28.7 us per layer, slower.)  fp6 operands are six dwords: dwords 6 and 7
                // of the builtin's eight are not read, so nothing is copied for them.
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int ks = n >> 1, i = n & 1;
                    if constexpr ((DBG & 8) != 0) {
                        keep_alive(F.a[ks][i]);
                        keep_alive(F.b[ks][0]);
                    }
                    else
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[ks][i], F.b[ks][0], acc[i][0], 0, 0, 0);
                    if (side_on && !(DBG & 4)) {   // (DBG 4, lab: no fragment reads behind the first K-tile's -- stale registers, timing only)
#pragma unroll
                        for (int p = 0; p < N_SIDE; ++p)
                            if (p >= n * N_SIDE / 4 && p < (n + 1) * N_SIDE / 4)
                                side_read(SIDE, side_kt, p);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const v8i av0 = v8i{(int)qa[0], (int)qa[1], (int)qa[2], (int)F.ra[0].x, (int)F.ra[0].y, (int)F.ra[0].z, (int)F.ra[0].w, (int)F.ra[0].w};
                const v8i av1 = v8i{(int)qa[3], (int)qa[4], (int)qa[5], (int)F.ra[1].x, (int)F.ra[1].y, (int)F.ra[1].z, (int)F.ra[1].w, (int)F.ra[1].w};
                const v8i bv0 = v8i{(int)F.rb[0].x, (int)F.rb[0].y, (int)F.rb[0].z, (int)qb[3], (int)qb[4], (int)qb[5], (int)qb[5], (int)qb[5]};
                if constexpr ((DBG & 8) != 0) {
                    keep_alive(av0);
                    keep_alive(av1);
                    keep_alive(bv0);
                }
                else {
                    acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av0, bv0, acc[0][0], 2, 2, 0, (int)F.ra[0].w, 0, (int)F.rb[0].w);
                    acc[1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av1, bv0, acc[1][0], 2, 2, 0, (int)F.ra[1].w, 0, (int)F.rb[0].w);
                }
                // all six dwords of both conversions stay allocated until here: the unused half of the frames' conversion was handed
                // out as a scratch register right behind its issue, and a vector instruction that WRITES a register of a conversion in
                // flight waits for it (seen in the ISA: v_add_u32 v10 behind v_cvt ... v[10:15])
                asm volatile("" ::"v"(qa), "v"(qb));
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // Skewed wave groups (8-wave tiles; waves w and w + 4 share a SIMD): between two barriers the EARLY wave of a SIMD reads
        // K-tile kt into registers and then issues its products, the LATE wave first issues the products of K-tile kt - 1 -- read in
        // the previous period -- and reads K-tile kt afterwards.  One wave of every SIMD feeds the matrix pipe while the other one
        // waits for LDS; with all eight waves in phase (reads, then products, every period) the LDS round trips and the matrix work
        // added up (ablations of profiles/r04/gemm_mx_ablation.log: 2.20 ms full, 1.20 ms reads + conversions alone, 1.19 ms of
        // matrix work).  Both groups touch stage kt only in period kt, so the ring and its one barrier per K-tile stay as they were;
        // every accumulator still sums its K-tiles in ascending order (bit-identical results).
        const bool late = (C::SKEW != ((DBG & 256) != 0)) && C::NW == 8 && wave >= C::NW / 2;
        // sync(kt): K-tile kt has landed in its stage for every wave; the stage of K-tile kt - 1 is free (every wave finished its reads
        // before it arrived here) and receives K-tile kt + STAGES - 1
        auto stamp = [&](int kt, int phase) {
#ifdef AMX_LAB
            if constexpr ((DBG & 2048) != 0) {
                if (blockIdx.x == 0 && vi == (int)blockIdx.x && kt < 48 && lane == 0 && (!LAST || n_tiles_total > 1000))
                    mx_stamps[(wave * 48 + kt) * 4 + phase] = __builtin_amdgcn_s_memtime();
            }
#endif
        };
        auto sync = [&](int kt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int ahead = min(C::STAGES - 2, KT - 1 - kt);  // K-tiles that stay in flight
            // `ahead` K-tiles stay in flight behind the awaited one: the wave's own operations per refill x ahead (+ for a prefetching
            // wave the prefetch load issued behind the awaited K-tile's pieces)
            if (C::PF_FREE && !issuer) {
                // no DMA of its own, and its prefetch loads are never waited for inside the loop
            }
            else if (C::PF > 0 && pf_wave)
                mx_wait_ahead<C::PPW + 1, 1, C::STAGES - 2>(ahead);
            else if (!issuer)
                mx_wait<0>();  // nothing of its own in flight: the issuing waves' waits + the barrier cover the K-tile
            else if (hi_wave)
                mx_wait_ahead<C::PPW, 0, C::STAGES - 2>(ahead);
            else
                mx_wait_ahead<C::PPW - 1, 0, C::STAGES - 2>(ahead);
            __builtin_amdgcn_s_barrier();
            stamp(kt, 0);
            if (kt + C::STAGES - 1 < KT && !(DBG & 16)) {
                dma_kt = kt + C::STAGES - 1;  // issued by refill() or, piece by piece, by the next products() (SPREAD)
            }
        };
        // The refill burst goes BEHIND the fragment reads of the K-tile: a piece holds the issuing wave until the address unit has
        // taken its 1 KB (eight waves share the unit: 360-800 cycles for a wave's six pieces, tools/mx_timeline.py), and the LDS
        // reads issued in front of it complete meanwhile -- issued behind it they added their ~500 cycles of latency to every period.
        auto refill = [&](int kt) {
            if constexpr (C::SPREAD == ((DBG & 512) != 0)) {
                if (dma_kt >= 0)
                    stage(dma_kt % C::STAGES, dma_kt);
                dma_kt = -1;
            }
            stamp(kt, 1);
        };
        // one code path for both groups -- early: sync(kt) reads(kt) products(kt); late: reads(kt) sync(kt + 1) products(kt), i.e. the
        // late wave's products of K-tile kt run in period kt + 1, in front of its reads of K-tile kt + 1.  Both execute KT barriers.
        if constexpr (C::PIPE == 2) {
            static_assert(DBG == 0 || DBG == 2048, "the self-pipelined K loop has no ablation variants");
            static_assert(C::NHI == 0 && C::IW == C::NW && C::MI == 4 && C::MJ == 4, "every wave issues PPW pieces");
            // ONE WAVE PER SIMD, pipelining itself (tuning tile=9; opt-in -- see the measurement below).  A K-tile's 48 matrix instructions
            // go out in three phases of 16 -- f16 products of k-slab 0 (A), of k-slab 1 (B), the scaled cross products (C) -- and everything
            // else rides between them, in their shadow:
            //   A  the four conversions of THIS K-tile (fragments and records are in registers from the previous iteration), their lane
            //      swaps and the assembly of the scaled operands;
            //   B  the fragment reads of k-slab 0 of the NEXT K-tile, into the registers phase A has just finished with, and eight of
            //      this wave's twelve LDS-DMA pieces of K-tile kt + 3 (into the stage of K-tile kt, which every wave has finished reading:
            //      the barrier at the top);
            //   C  the reads of k-slab 1 and of the records of the next K-tile, the other four pieces.
            // One barrier per K-tile, at the top: this wave's reads of K-tile kt are complete (lgkmcnt), its pieces of K-tile kt + 1
            // have landed (counted vmcnt, K-tile kt + 2 stays in flight).  Per accumulator the same three instructions in the same
            // order as in every other configuration: bit-identical.  sched_barrier(0) pins the interleave (left to itself the
            // scheduler gathers the reads at the head of the block and the matrix instructions behind them).  180-186 VGPRs + 256 AGPRs,
            // no scratch (the epilogues pin the accumulator blocks in AGPRs until their use: pin_block).
            // MEASURED (one box, output layer 2048 -> 10000): 1.99 ms against the ping-pong tile's 1.87 ms.  With neither reads nor pieces
            // the loop runs 1.41 ms; the twelve pieces cost 0.35 ms, the 24 reads 0.21 ms (additive; all operands L2 hits: -0.1 ms) --
            // a wave issues in order, and what is not a matrix instruction takes issue time from the one wave a SIMD has, where the
            // ping-pong tile gives it to the partner wave.  And the comparison is not run at one clock: BOTH kernels sit at the
            // package's 1400 W power cap (profiles/r05/power_probe.log) -- the ping-pong tile at 1.88 GHz, this one at 2.03-2.1 GHz,
            // round 4's in-phase tile at 2.08 GHz: the firmware trades clock for every gain in instructions per cycle.
            auto await_tile = [&](int ahead_tiles) { mx_wait_ahead<C::PPW, 0, 2>(ahead_tiles); };
            const unsigned vdma = (unsigned)lane * 16u;
            auto dma = [&](int q, int slot, int kt) {  // piece q of this wave, unconditionally
                const char*    src = (p_b[q] ? xblk : wblk) + (size_t)kt * BLK + p_src[q];
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + slot * C::STAGE_BYTES + p_dst[q]));
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vdma), "s"(src), "s"(dst) : "memory");
            };
            Frag& F = fr[0];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            await_tile(min(2, KT - 1));
            __builtin_amdgcn_s_barrier();
            reads(0, F);
            auto f16_products = [&](int ks, int i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[ks][i], F.b[ks][j], acc[i][j], 0, 0, 0);
            };
            auto frame_swaps = [&](int j, u32x4 (&x0)[2], u32x4 (&x1)[2]) {  // see products(): frame-major copies of the pair's fragments
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    x0[ks] = __builtin_bit_cast(u32x4, F.b[ks][j]);
                    x1[ks] = __builtin_bit_cast(u32x4, F.b[ks][j + 1]);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const u32x2 sw = __builtin_amdgcn_permlane32_swap(x0[ks][d], x1[ks][d], false, false);
                        x0[ks][d]      = sw[0];
                        x1[ks][d]      = sw[1];
                    }
                }
            };
            auto frame_convert = [&](int j, u32x4 (&x0)[2], u32x4 (&x1)[2]) {
                const unsigned sc = (fk ? F.rb[j + 1].w : F.rb[j].w) + 11u;
                return q_fields_pair(__builtin_bit_cast(f16x8, x0[0]), __builtin_bit_cast(f16x8, x0[1]), __builtin_bit_cast(f16x8, x1[0]),
                                     __builtin_bit_cast(f16x8, x1[1]), sc);
            };
            v8i      av[4], bv[4];
            unsigned sa[4], sb[4];
            auto frame_operands = [&](int j, u32x6 q) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const u32x2 sw = __builtin_amdgcn_permlane32_swap(q[d], q[3 + d], false, false);
                    q[d]           = sw[0];
                    q[3 + d]       = sw[1];
                }
                bv[j]     = v8i{(int)F.rb[j].x, (int)F.rb[j].y, (int)F.rb[j].z, (int)q[0], (int)q[1], (int)q[2], 0, 0};
                bv[j + 1] = v8i{(int)F.rb[j + 1].x, (int)F.rb[j + 1].y, (int)F.rb[j + 1].z, (int)q[3], (int)q[4], (int)q[5], 0, 0};
                sb[j]     = F.rb[j].w;
                sb[j + 1] = F.rb[j + 1].w;
            };
            auto row_operands = [&](int i, const u32x6& q) {
                av[i]     = v8i{(int)q[0], (int)q[1], (int)q[2], (int)F.ra[i].x, (int)F.ra[i].y, (int)F.ra[i].z, 0, 0};
                av[i + 1] = v8i{(int)q[3], (int)q[4], (int)q[5], (int)F.ra[i + 1].x, (int)F.ra[i + 1].y, (int)F.ra[i + 1].z, 0, 0};
                sa[i]     = F.ra[i].w;
                sa[i + 1] = F.ra[i + 1].w;
            };
            auto iteration = [&](int kt, auto with_dma) {
                constexpr bool DMA = decltype(with_dma)::value;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (kt + 1 < KT)
                    await_tile(kt + 2 < KT ? 1 : 0);
                __builtin_amdgcn_s_barrier();
                stamp(kt, 0);
                __builtin_amdgcn_sched_barrier(0);
                const char* ab   = lds + ((kt + 1) % C::STAGES) * C::STAGE_BYTES;  // the next K-tile's stage (the last iteration reads a stale one)
                const char* bb   = ab + C::A_BYTES;
                const int   slot = kt % C::STAGES;                                  // free: refilled with K-tile kt + 3
                // side operation n of the K-tile: 0-23 the fragment / record reads of the NEXT K-tile (k-slab 0 first: its registers are
                // free since phase A), 24-35 this wave's LDS-DMA pieces.  ONE of them behind a matrix instruction (a wave issues in order:
                // two pieces in a row hold it ~70 cycles -- the first form of this loop, 2.07 ms -- one mostly fits into the time the matrix
                // pipe is busy with the instruction in front of it: 1.99 ms); a piece behind every fourth instruction of all three phases
                // and a per-wave skew of 16 cycles were measured too, both slower (profiles/r05/one_wave_per_simd.log)
                auto side = [&](int n) {
                    if (n < 4)
                        F.a[0][n] = *(const f16x8*)(ab + h_off(a_row + 32 * n, fk));
                    else if (n < 8)
                        F.b[0][n - 4] = *(const f16x8*)(bb + h_off(b_row + 32 * (n - 4), fk));
                    else if (n < 12)
                        F.a[1][n - 8] = *(const f16x8*)(ab + h_off(a_row + 32 * (n - 8), 2 + fk));
                    else if (n < 16)
                        F.b[1][n - 12] = *(const f16x8*)(bb + h_off(b_row + 32 * (n - 12), 2 + fk));
                    else if (n < 20)
                        F.ra[n - 16] = __builtin_bit_cast(uint4, *(const f16x8*)(ab + C::A_R + fk * (C::BN * 16) + (a_row + 32 * (n - 16)) * 16));
                    else if (n < 24)
                        F.rb[n - 20] = __builtin_bit_cast(uint4, *(const f16x8*)(bb + C::B_R + fk * (C::BT * 16) + (b_row + 32 * (n - 20)) * 16));
                    else if (n < 24 + C::PPW) {
                        if constexpr (DMA)
                            dma(n - 24, slot, kt + 3);
                    }
                };
                // ---- A
                u32x4 x0[2], x1[2];
                frame_swaps(0, x0, x1);
                f16_products(0, 0);
                __builtin_amdgcn_sched_barrier(0);
                u32x6 qb = frame_convert(0, x0, x1);
                f16_products(0, 1);
                __builtin_amdgcn_sched_barrier(0);
                frame_operands(0, qb);
                frame_swaps(2, x0, x1);
                f16_products(0, 2);
                __builtin_amdgcn_sched_barrier(0);
                qb       = frame_convert(2, x0, x1);
                u32x6 qa = q_fields_pair(F.a[0][0], F.a[1][0], F.a[0][1], F.a[1][1], F.ra[0].w);
                f16_products(0, 3);
                __builtin_amdgcn_sched_barrier(0);
                frame_operands(2, qb);
                row_operands(0, qa);
                qa = q_fields_pair(F.a[0][2], F.a[1][2], F.a[0][3], F.a[1][3], F.ra[2].w);
                // ---- B
                stamp(kt, 1);
#pragma unroll
                for (int m = 0; m < 16; ++m) {  // reads 0-7 behind the first eight, pieces 0-7 behind the rest
                    acc[m / 4][m % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a[1][m / 4], F.b[1][m % 4], acc[m / 4][m % 4], 0, 0, 0);
                    if (m == 0)
                        row_operands(2, qa);
                    side(m < 8 ? m : 24 + (m - 8));
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- C
                stamp(kt, 2);
#pragma unroll
                for (int m = 0; m < 16; ++m) {  // reads 8-23 behind the first twelve (two behind each of the first four), pieces 8-11 behind the rest
                    acc[m / 4][m % 4] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av[m / 4], bv[m % 4], acc[m / 4][m % 4], 2, 2, 0, (int)sa[m / 4], 0, (int)sb[m % 4]);
                    if (m < 4) {
                        side(8 + 2 * m);
                        side(9 + 2 * m);
                    }
                    else if (m < 12)
                        side(12 + m);
                    else
                        side(24 + 8 + (m - 12));
                    __builtin_amdgcn_sched_barrier(0);
                }
                stamp(kt, 3);
            };
            int kt = 0;
            for (; kt + 3 < KT; ++kt)
                iteration(kt, std::true_type{});
#pragma unroll 1
            for (; kt < KT; ++kt)
                iteration(kt, std::false_type{});
        }
        else if constexpr (C::PIPE == 1) {
            static_assert(DBG == 0 || DBG == 2048 || DBG == 128, "the ping-pong K loop: time stamps, and the no-conversion ablation (lab builds)");
            const bool second = wave >= C::NW / 2;  // the group that runs half a period behind (waves 4-7: the partners of waves 0-3)
            auto await_tile = [&](int ahead_tiles, int max_ahead) {  // own pieces of a K-tile have landed; `ahead_tiles` younger K-tiles may stay in flight
                if (!issuer)
                    mx_wait<0>();
                else if (max_ahead >= 2) {
                    if (hi_wave)
                        mx_wait_ahead<C::PPW, 0, 2>(ahead_tiles);
                    else
                        mx_wait_ahead<C::PPW - 1, 0, 2>(ahead_tiles);
                }
                else {
                    if (hi_wave)
                        mx_wait_ahead<C::PPW, 0, 1>(ahead_tiles);
                    else
                        mx_wait_ahead<C::PPW - 1, 0, 1>(ahead_tiles);
                }
            };
            // prologue: K-tiles 0 .. 2 are on their way (all three stages); K-tile 0 has landed -> the first group's image
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            await_tile(min(2, KT - 1), 2);
            __builtin_amdgcn_s_barrier();
            // ONE loop body for both groups -- barrier, load K-tile kt (refill share + fragment reads), barrier, products of K-tile kt --
            // and the first group runs it one barrier AHEAD of the second: it skips the first barrier of the loop (and executes one
            // more behind it), so its load segments coincide with its partners' compute segments and the other way round.  (Branching
            // the roles inside the loop, or one loop per group, made the register allocator keep copies of the accumulators: 170-700
            // bytes of scratch per lane.)  With b0, b1, ... the barriers behind the prologue's, the first group loads K-tile kt in
            // [b(2 kt - 1), b(2 kt)] and the second in [b(2 kt), b(2 kt + 1)]:
            //   * the refill issued in load(kt) is K-tile kt + 2 -> the stage of K-tile kt - 1, which
            //     the first group read in [b(2 kt - 3), b(2 kt - 2)] and the second in [b(2 kt - 2), b(2 kt - 1)]: free for both;
            //   * every wave has awaited its pieces of K-tile kt in front of b(2 kt - 1) at the latest (the await of load(kt - 1)).
            for (int kt = 0; kt < KT; ++kt) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (second || kt > 0)
                    __builtin_amdgcn_s_barrier();
                stamp(kt, 0);
                // (measured on one box, profiles/r05/pingpong_ab.log: the fragment reads in front of the refill's pieces +0.7 %; a static
                // s_setprio 1 for the second group: no change; the conversions moved to the end of the load segment, so that the compute
                // segment is matrix instructions only: output layer 1.86 -> 3.0 ms -- a conversion does not run beside the partner's
                // matrix instructions (tools/cvt_rate.hip), it takes their pipe)
                if (kt >= 1 && kt + 2 < KT)
                    stage((kt + 2) % C::STAGES, kt + 2);
                reads(kt, fr[0]);
                stamp(kt, 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (kt + 1 < KT)
                    await_tile(kt + 2 < KT ? 1 : 0, 1);  // own pieces of K-tile kt + 1 have landed; kt + 2 may stay in flight
                __builtin_amdgcn_s_barrier();
                stamp(kt, 2);
                products(fr[0]);
                stamp(kt, 3);
            }
            if (!second)
                __builtin_amdgcn_s_barrier();
        }
        else if constexpr (C::PIPE == 3) {
            static_assert(DBG == 0 || DBG == 2048 || DBG == 4 || DBG == 8 || DBG == 16 || DBG == 24 || DBG == 28 || DBG == 20 || DBG == 128 || DBG == 156, "read-ahead K loop: time stamps and the ablations");
            static_assert(C::U == 2, "two register images that swap roles every K-tile: an even number of K-tiles per barrier keeps the roles fixed");
            // K-tiles up to `needed` have landed for this wave's own pieces; issued so far: up to `issued` (both clamped to the last K-tile)
            auto await = [&](int needed, int issued) {
                const int ahead = max(0, min(issued, KT - 1) - min(needed, KT - 1));
                if (!issuer)
                    mx_wait<0>();
                else if (hi_wave)
                    mx_wait_ahead<C::PPW, 0, C::STAGES - C::U>(ahead);
                else
                    mx_wait_ahead<C::PPW - 1, 0, C::STAGES - C::U>(ahead);
            };
            // TWO register images: while the products of K-tile k are issued from one, the reads of K-tile k + 1 fill the other, one or
            // two reads behind every matrix instruction (products_side).  The ring keeps its schedule -- barrier i frees the slots of
            // iteration i - 1 (K-tile 2 i - 2 was read under the products of 2 i - 3, K-tile 2 i - 1 under those of 2 i - 2; every
            // wave awaits its reads in front of the barrier) and the loaders refill them -- with one more K-tile awaited: the second
            // half of iteration i reads K-tile 2 i + 2, the first of iteration i + 1.
            __builtin_amdgcn_s_waitcnt(0xc07f);
            await(0, C::STAGES - C::U - 1);  // K-tile 0 has landed
            __builtin_amdgcn_s_barrier();
            if (!loader)
                reads(0, fr[0]);
            const int n_it = (KT + C::U - 1) / C::U;
            // STEADY iterations (kt + STAGES - 1 < KT: both K-tiles of the refill exist, nothing is clamped) know every count at
            // compile time: ONE s_waitcnt with a constant immediate and no guards.  The general form -- the nested selection of the
            // immediate from a run-time `ahead`, the `< KT` guards of every read and piece -- compiled to ~60 scalar instructions and
            // ~25 branches per iteration, and the EMPTY loop (no reads, no matrix instructions, no DMA, no conversions) took 0.36 us
            // per iteration, half of the full loop's time (profiles/r06/hidden_layer_probe.log); it now runs for the tail only.
            auto body = [&](int it, auto steady_tag) {
                constexpr bool STEADY = decltype(steady_tag)::value;
                const int kt = it * C::U;
                // this wave's reads so far are complete.  The BUILTIN, not an asm statement: the compiler's wait-count pass must know
                // that the image read under the previous products has arrived -- behind an asm wait it does not, and protects the
                // first product with an s_waitcnt lgkmcnt of its own, which (LDS returns in order) waits for the reads just issued
                // for the NEXT K-tile: the overlap this loop exists for was gone (seen in the ISA)
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0), vmcnt / expcnt untouched
                // K-tiles kt + 1 .. kt + U have landed (own pieces); K-tiles up to kt + STAGES - U - 1 are issued
                if constexpr (STEADY) {
                    constexpr int AHEAD = C::STAGES - 2 * C::U - 1;
                    if (issuer) {   // (a computing wave has no vector-memory operation in flight inside the loop)
                        if (hi_wave)
                            mx_wait<AHEAD * C::PPW>();
                        else
                            mx_wait<AHEAD * (C::PPW - 1)>();
                    }
                }
                else
                    await(kt + C::U, kt + C::STAGES - C::U - 1);
                __builtin_amdgcn_s_barrier();                   // ... for every wave; the slots of iteration it - 1 are free
                stamp(it, 0);
                if (issuer && !(DBG & 16)) {
#pragma unroll
                    for (int u = 0; u < C::U; ++u) {
                        const int n = kt + u + C::STAGES - C::U;
                        if (STEADY || n < KT)
                            stage(n % C::STAGES, n);
                    }
                }
                stamp(it, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (!loader) {
#pragma unroll
                    for (int u = 0; u < C::U; ++u)
                        if (STEADY || kt + u < KT)
                            products_ahead(fr[u & 1], fr[(u + 1) & 1], kt + u + 1, STEADY || kt + u + 1 < KT);
                    // the products stay in front of the next iteration's waits: a matrix instruction touches no memory, so neither
                    // the waits nor the barrier hold it -- left alone the compiler sank all of them behind the next top-of-loop
                    // s_waitcnt lgkmcnt(0) (seen in the ISA)
#pragma unroll
                    for (int i = 0; i < C::MI; ++i)
#pragma unroll
                        for (int j = 0; j < C::MJ; ++j)
                            asm volatile("" : "+v"(acc[i][j]));
                }
                __builtin_amdgcn_sched_barrier(0);
                stamp(it, 3);
            };
            int it = 0;
            for (; it * C::U + C::STAGES - 1 < KT; ++it)
                body(it, std::true_type{});
#pragma unroll 1
            for (; it < n_it; ++it)
                body(it, std::false_type{});
        }
        else if constexpr (C::U > 1 || C::LW > 0) {
            static_assert(!C::SKEW && !C::SPREAD && C::PF == 0 && (C::IW == C::NW || C::LW > 0), "plain burst refill only");
            for (int kt = 0; kt < KT; kt += C::U) {
                const int nk = min(C::U, KT - kt);
                // K-tiles kt .. kt + nk - 1 have landed; issued so far: up to kt + STAGES - U - 1
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int ahead = max(0, min(kt + C::STAGES - C::U - 1, KT - 1) - (kt + nk - 1));
                if (!issuer)
                    mx_wait<0>();
                else if (hi_wave)
                    mx_wait_ahead<C::PPW, 0, C::STAGES - C::U - 1>(ahead);
                else
                    mx_wait_ahead<C::PPW - 1, 0, C::STAGES - C::U - 1>(ahead);
                __builtin_amdgcn_s_barrier();  // ... for every wave, and the stages read in the previous iteration are free
                stamp(kt / C::U, 0);
                if (!loader) {
#pragma unroll
                    for (int u = 0; u < C::U; ++u)
                        if (u < nk)
                            reads(kt + u, fr[u]);
                }
                if (issuer) {
#pragma unroll
                    for (int u = 0; u < C::U; ++u) {
                        const int n = kt + u + C::STAGES - C::U;
                        if (n < KT)
                            stage(n % C::STAGES, n);
                    }
                }
                stamp(kt / C::U, 1);
                if constexpr ((DBG & 2048) != 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    stamp(kt / C::U, 2);
                }
                if (!loader) {
#pragma unroll
                    for (int u = 0; u < C::U; ++u)
                        if (u < nk)
                            products(fr[u]);
                }
                stamp(kt / C::U, 3);
            }
        }
        else {
        if (late) {
            sync(0);
            if (dma_kt >= 0)  // the refill that belongs to barrier 0: the late wave has no products to spread it over yet
                stage(dma_kt % C::STAGES, dma_kt);
            dma_kt = -1;
        }
        for (int kt = 0; kt < KT; ++kt) {
            if (!late)
                sync(kt);
            reads(kt, fr[0]);
            if (late && kt + 1 < KT)
                sync(kt + 1);
            if constexpr ((DBG & 4096) != 0)  // ablation: the burst in front of the reads' completion, as before
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            refill(kt);
            if constexpr ((DBG & 2048) != 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                stamp(kt, 2);
            }
            products(fr[0]);
            stamp(kt, 3);
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (C::PF > 0)
            asm volatile("" ::"v"(pf_reg));
        tile_stamp(1);
        if constexpr (KSPLIT) {
            constexpr int PER_WAVE = C::MI * C::MJ * 16 * 64;  // [tile][group][wave][register][lane]: 256 contiguous bytes per wave-instruction
            if (ks > 1) {  // launch 1: park the partial sums, no epilogue
                float* mine = ks_ws + ((size_t)(vi * ks + kgroup) * C::NW + (wave % C::NW)) * PER_WAVE + lane;
                if (!loader) {
#pragma unroll
                    for (int i = 0; i < C::MI; ++i)
#pragma unroll
                        for (int j = 0; j < C::MJ; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                mine[((i * C::MJ + j) * 16 + r) * 64] = acc[i][j][r];
                }
                __syncthreads();
                KT = KT_all;
                continue;
            }
            if (reduce && !loader) {  // launch 2: ((P0 + P1) + P2) + P3
                const int ng = -ksplit;
#pragma unroll 1
                for (int g = 0; g < ng; ++g) {
                    const float* theirs = ks_ws + ((size_t)(vi * ng + g) * C::NW + (wave % C::NW)) * PER_WAVE + lane;
#pragma unroll
                    for (int i = 0; i < C::MI; ++i)
#pragma unroll
                        for (int j = 0; j < C::MJ; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float p = theirs[((i * C::MJ + j) * 16 + r) * 64];
                                acc[i][j][r]  = g == 0 ? p : acc[i][j][r] + p;
                            }
                }
            }
        }
        // the epilogue's lane-dependent addresses are derived from opaque copies of the lane / thread id: computed from `lane` they are
        // invariants of the tile loop, get hoisted in front of the K-loop and spilled there (the K-loop owns the register file) -- and a
        // scratch access inside the loop would join the queue the counted vmcnt waits count
        int elane = lane, etid = tid;
        asm volatile("" : "+v"(elane), "+v"(etid));
        static_assert(!(LAST && C::LW > 0), "loader waves: hidden layers only (gemm_epilogue's barriers expect NW computing waves)");
        if (LAST)
            gemm_epilogue<C, ACT, true>(acc, lds, s_bias, out, ldo, 0, n_valid, t_valid, n0, t0, tile_n, wn, wt, elane, etid, part_min, part_idx, part_ld);
        else {
            // next layer's blocks, straight from the accumulators: lane (frame, hh) of result block (i, j) owns natural indices
            // 8 g + 4 hh + e of K-tile (n0 + wn WNR + 32 i) / 32, i.e. half hh of that row
            __syncthreads();  // bias visible
            const int tl32 = elane & 31, hh = elane >> 5;
            bool      over = false;
            if (!loader) {
#pragma unroll
            for (int j = 0; j < C::MJ; ++j) {
                const int t = t0 + wt * WTT + 32 * j + tl32;
#pragma unroll
                for (int i = 0; i < C::MI; ++i) {
                    pin_block<C>(acc[i][j]);
                    float v[16], m = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b4 = *(const float4*)(s_bias + wn * WNR + 32 * i + 8 * g + 4 * hh);
                        v[4 * g + 0]    = activate<ACT>(acc[i][j][4 * g + 0] + b4.x);
                        v[4 * g + 1]    = activate<ACT>(acc[i][j][4 * g + 1] + b4.y);
                        v[4 * g + 2]    = activate<ACT>(acc[i][j][4 * g + 2] + b4.z);
                        v[4 * g + 3]    = activate<ACT>(acc[i][j][4 * g + 3] + b4.w);
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        m = (v[u] != v[u]) ? __builtin_inff() : fmaxf(m, fabsf(v[u]));  // NaN: out of range
                    m    = fmaxf(m, __shfl_xor(m, 32, 64));
                    over = over || !(m < 65520.f);
                    const int ec = block_exponent(m);
                    char*     blk = (char*)out + ((size_t)(t >> 8) * ktn + ((n0 + wn * WNR + 32 * i) >> 5)) * BLK;
                    lane_store(blk, t & 255, hh, lane_pack(v, ec, ec - 13));
                }
            }
            }
            if (over)
                *overflow = 1u;
        }
        __syncthreads();  // LDS (stages / epilogue scratch / bias) is reused by the next tile
        tile_stamp(2);
        KT = KT_all;
    }
}

}  // namespace mx
}  // namespace amx
