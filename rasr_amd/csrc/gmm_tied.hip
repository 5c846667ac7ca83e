// gmm_tied.hip -- pruned exact maximum-approximation scorer for tied models whose mixtures all list the same densities
// (Mm::GaussDiagonalMaximumFeatureScorer on a tied-mixture set: Mm/GaussDiagonalMaximumFeatureScorer.cc:116-141).
//
// score(t, m) = min_k s_k with s_k = (m2lw[m][k] + logNorm[k]) + dist[k][t] in f64 and the sequential rule
// `if ((double)best > s) { best = (float)s; idx = k; }`.  gmm_tied_tile_kernel (gmm.hip) evaluates the (min,+) product
// densely: K x n_mix x T sums, twice.  But a density far from the frame cannot win in ANY mixture, and that can be proven
// per (density, frame, 64-mixture tile) from small tables:
//
//   a^[k][m]     = fl32(m2lw[k][m] + logNorm[k])             (model; the bound kernel reads a bf16 image rounded up, [K][mix_pad])
//   amin[j][k]   = min over the mixtures of tile j of a^[k][m] (model, [n_tiles][Kpad]);  aminG[k] = min over all tiles
//   U[t][m]      = min over 64 densities NEAR frame t (the closest density of each residue class k mod 64; 32 until round 6) of
//                  s^_k = fl32(a^[k][m] + dist[k][t])         -- an upper bound of min_k s^_k, because it is a minimum over a subset
//
// Candidates -- the densities that can influence (best, idx), i.e. those whose f64 sum rounds to the winning f32 value, see
// the subsequence argument in front of gmm_tied_tile_kernel -- satisfy s^_k <= min s^ + tau <= U[t][m] + tau', with
// tau' = 2^-21 (2 max_k|a^[.][m]| + |U|) (the bound of that comment, written for an upper bound of the minimum: |min s^| <=
// max(|U|, max|a^|) since dist >= 0).  With Thr[t][j] = max over the tile's mixtures of U + tau', and fl32 monotone,
//   fl32(amin[j][k] + dist[k][t]) > Thr[t][j]   ==>   no mixture of tile j has density k as a candidate for frame t
// and the same with aminG and ThrG[t] = max_j Thr[t][j] for the whole model.  The kernels:
//   gmm_dist_list_kernel    (gmm.hip; lists that name every density at most once, one pass) the frame-major distances dt[t][k] in
//                           list order AND the frame's near densities: per residue class k mod 64 the smallest 64-bit key (distance
//                           bits, position), kept with atomic minima; tied_list_kernel puts the keys back into their empty state
//   tied_transpose_kernel + tied_near_kernel   the same two results for every other case (a list that repeats a density, a call cut
//                           into passes, a dimension without an instance, builds with another number of near densities), from
//                           gmm_dist_kernel's density-major distances
//   tied_bound8_kernel      U and the mixtures' thresholds U + tau', Thr[t][j]: eight mixtures per lane, the near rows of a bf16 image of
//                           a^ rounded up in 1 KB requests, table segments with a home XCD (tied_bound_kernel: tables of 4 GB and more)
//   tied_list_kernel        per frame: the densities that pass the model-wide test (4 % on the config-3 instance), ascending, with
//                           their distances and log-normalisation terms
//   tied_mask_kernel        per (frame, tile): which list entries pass the TILE's test (1.3 % of all densities), as bit masks
//   tied_pruned_kernel      per (tile, frame): the survivors compacted into LDS, each lane's own f32 screen over them (1.15 of ~54
//                           pass per mixture), and the reference's f64 rule over what is left, in ascending k.
// Running the rule over a subsequence that contains every candidate, in the original order, is bit-identical.
//
// What bounds these kernels (round 6's measurements, profiles/r06/tied_steps.log): the pruned kernel is 40 192 one-wave items of very
// different length (16 .. 180 survivors) at eight waves per SIMD -- neither 12 % fewer vector instructions nor one round trip less per
// wave nor persistent waves move its 48 us; the bound kernel is its 329 MB of row requests (22 us with every row an L2 hit); the small
// kernels are as long as their chain of memory round trips, hence one trip per wave in the list / mask kernels.
// (profiles/r03/NOTES_tied.md has the measurements that led to the 6-instruction survivor test.)
//
// A model / feature distribution without that structure (everything survives) would make this slower than the dense
// kernel -- each table element is then used once instead of 16 times from registers.  The kernel counts the survivors; the host
// reads the count of earlier calls and goes back to gmm_tied_tile_kernel while the surviving fraction is high (gmm.hip).
#include "common.hpp"

#include <cfloat>
#include <cstring>
#include <vector>

#ifndef AMX_TIED_EXP
#define AMX_TIED_EXP 0  // 4: histograms of candidates per mixture and survivors per (frame, tile) on stderr after 20 calls
#endif

namespace amx {
#if AMX_TIED_EXP & 4
__device__ unsigned long long g_tied_hist[64];  // [0..31] candidates per (lane, wave); [32..63] survivors / 8 per wave
#endif

#ifndef AMX_TIED_NEAR
#define AMX_TIED_NEAR 64  // round 6 (profiles/r06/tied_ab.log): 32 until then.  Tighter bounds U leave 1.33 % instead of 1.59 % of the (density, frame,
                          // tile) triples to the pruned scorer -- 74 -> 48 us -- for twice the bound kernel's row reads (17 -> 31 us): 0.140 -> 0.127 ms
                          // per 256 frames; 16: 0.240 ms; 128: see the log.  Measurement builds: -DAMX_TIED_NEAR=16 | 32 | 128
#endif
constexpr int kTiedNear     = AMX_TIED_NEAR;   // near densities per frame = residue classes of the density index
constexpr int kTiedCounters = 256;  // survivor counters (summed by the host)

// dist [n_dens][Tpad] (coalesced along frames) -> dt [T][Kpad] (coalesced along the density list); 64 x 64 tiles through LDS
// (g_dist points at the first column of the pass: cols = the padded columns from there on, Tpad = the row stride)
__global__ __launch_bounds__(256) void tied_transpose_kernel(const float* __restrict__ g_dist, const uint32_t* __restrict__ g_k_dens, int K,
                                                           int Kpad, int T, int cols, int Tpad, float* __restrict__ g_dt) {
    __shared__ float s[64][65];
    const int        k0 = blockIdx.x * 64, t0 = blockIdx.y * 64;
    const int        c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = k0 + r0 + 4 * i;
        s[r0 + 4 * i][c] = (k < K && t0 + c < cols) ? g_dist[(size_t)g_k_dens[k] * Tpad + t0 + c] : __builtin_inff();
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int t = t0 + r0 + 4 * i;
        if (t < T)
            g_dt[(size_t)t * Kpad + k0 + c] = s[c][r0 + 4 * i];  // k >= K: +inf
    }
}

// The frame's closest density of every residue class k mod 32, for tied_bound_kernel (any subset gives a valid bound; this one needs
// no selection).  One workgroup per frame; nd / nk [T][32].
constexpr int kTiedNearThreads = 1024;
static_assert(kTiedNearThreads % kTiedNear == 0 && kTiedNear >= 8 && kTiedNear <= 128, "a thread of tied_near_kernel stays inside one residue class");

// A near density is ONE 64-bit key, (distance bits << 32) | list position: distances are >= +0, so the keys order like the distances
// (NaN above +inf), a minimum over keys is the closest density, and gmm_dist_list_kernel (gmm.hip) can keep the minima with atomics
// instead of this kernel.  kTiedNearInit = (+inf, row 0) is what an empty class holds (row 0 stands in, its sum is +inf).
constexpr unsigned long long kTiedNearInit = 0x7f80000000000000ull;

__global__ __launch_bounds__(256) void tied_near_init_kernel(unsigned long long* __restrict__ g_near, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        g_near[i] = kTiedNearInit;
}

__global__ __launch_bounds__(kTiedNearThreads) void tied_near_kernel(const float* __restrict__ g_dt, int K, int Kpad,
                                                                    unsigned long long* __restrict__ g_near) {
    constexpr int       NT = kTiedNearThreads;
    __shared__ float    s_v[NT];
    __shared__ uint32_t s_i[NT];
    const int           t = blockIdx.x, tid = threadIdx.x;
    const float*        row = g_dt + (size_t)t * Kpad;
    float               bv = __builtin_inff();
    uint32_t            bi = 0;
    for (int k = tid; k < K; k += NT) {  // NT is a multiple of 32: a thread stays inside one residue class
        const float v = row[k];
        if (v < bv) {
            bv = v;
            bi = (uint32_t)k;
        }
    }
    s_v[tid] = bv;
    s_i[tid] = bi;
    __syncthreads();
    if (tid < kTiedNear) {
        for (int j = 1; j < NT / kTiedNear; ++j)
            if (s_v[tid + kTiedNear * j] < bv) {
                bv = s_v[tid + kTiedNear * j];
                bi = s_i[tid + kTiedNear * j];
            }
        // +inf: empty class, or no finite distance (NaN / inf frame); row 0 stands in, its sum is +inf
        g_near[(size_t)t * kTiedNear + tid] = ((unsigned long long)__float_as_uint(bv) << 32) | bi;
    }
}

// Thr[t][tile] and the mixtures' own thresholds: U = min over the frame's near densities of fl32(a^ + dist); the tile's threshold is
// the maximum of U + tau' over its real mixtures.
// The sums only have to bound the minimum from ABOVE, so the 32 table rows per frame are read from a bf16 image of a^ that was
// rounded UP (a^_up >= a^, hence fl32(a^_up + dist) >= fl32(a^ + dist)): half the bytes of the kernel's only real traffic, for a
// bound that is at most 2^-8 |a^| looser.
// A lane takes TWO mixtures (one dword = two bf16 per row): the near densities and their distances are wave-uniform scalars, so a row
// costs the wave one load, two unpacks, two sums and two minima for 128 mixtures.  64 mixtures = one tile = 32 lanes.
__global__ __launch_bounds__(256) void tied_bound_kernel(const unsigned short* __restrict__ g_aup, const float* __restrict__ g_amax,
                                                        const uint2* __restrict__ g_near, int n_mix,
                                                        int mix_pad, int n_tiles, float* __restrict__ g_thr, float* __restrict__ g_thr_m,
                                                        int seg_per_xcd) {
    // Workgroup b runs on XCD b % 8, and each XCD has its own L2.  A near row is wanted by ~4 of a 256-frame batch's frames, so a 1 KB
    // segment of the table (512 mixtures) always goes to the SAME XCD, whatever the frame: segment s on XCD s % 8 (round 6; the grid is
    // 8 x segments-per-XCD x T, one-dimensional).  The table's 82 MB are then spread over the eight L2s instead of streamed through each.
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t = idx / seg_per_xcd, seg = xcd + 8 * (idx - t * seg_per_xcd);
    const int m = 2 * (seg * 256 + threadIdx.x);  // mix_pad is a multiple of 64: m + 1 < mix_pad with m
    if (m >= mix_pad)
        return;  // whole 32-lane halves leave together (a half = one tile)
    const uint2*    nr = g_near + (size_t)t * kTiedNear;  // .x = list position, .y = distance bits (little-endian halves of the key)
    uint32_t        v[kTiedNear];
#pragma unroll
    for (int i = 0; i < kTiedNear; ++i)  // all rows in flight before the first use
        v[i] = *(const uint32_t*)(g_aup + (size_t)nr[i].x * mix_pad + m);
    float u0 = FLT_MAX, u1 = FLT_MAX;
#pragma unroll
    for (int i = 0; i < kTiedNear; ++i) {
        const float d = __uint_as_float(nr[i].y);
        u0            = fminf(u0, __uint_as_float(v[i] << 16) + d);
        u1            = fminf(u1, __uint_as_float(v[i] & 0xffff0000u) + d);
    }
    // tau' = 2^-21 (2 max|a^| + |U|)
    float thr0 = -__builtin_inff(), thr1 = -__builtin_inff();
    if (m < n_mix) {
        thr0 = u0 + (4.76837158e-7f * (2.f * g_amax[m] + fabsf(u0)) + 1e-30f);
        if (!(thr0 == thr0))
            thr0 = __builtin_inff();
    }
    if (m + 1 < n_mix) {
        thr1 = u1 + (4.76837158e-7f * (2.f * g_amax[m + 1] + fabsf(u1)) + 1e-30f);
        if (!(thr1 == thr1))
            thr1 = __builtin_inff();
    }
    *(float2*)(g_thr_m + (size_t)t * mix_pad + m) = make_float2(thr0, thr1);  // the mixtures' own thresholds: tied_pruned_kernel screens with them (-inf: padding)
    float thr = fmaxf(thr0, thr1);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
        thr = fmaxf(thr, __shfl_xor(thr, o));
    const int tile = m >> 6;
    if ((lane & 31) == 0 && tile < n_tiles)
        g_thr[(size_t)t * n_tiles + tile] = thr;
}

// The same bounds with EIGHT mixtures per lane, for tables below 4 GB (round 6).  Measured on the kernel above: with every row an L2
// hit it still takes 25.7 of its 28 us, with fewer instructions but fewer waves per SIMD it is slower -- it is bound by the 20 096 x 64
// dword-per-lane loads, each a 256-byte wave request, which leave a CU at ~50 GB/s where 1 KB requests reach 128 GB/s
// (profiles/r06/l2_probe.log).  Here a wave reads 1 KB of a row per instruction (one wave = one 1 KB segment of the table = eight
// tiles, on XCD segment % 8 as above), a quarter of the load and scalar instructions per byte; rows arrive eight at a time into one of
// two register buffers while the other is summed (two packed adds per dword pair, one three-way minimum per two rows and mixture).
#ifndef AMX_TIED_BOUND_B
#define AMX_TIED_BOUND_B 4     // measured 4 x 2: 21.5 us, 2 x 2: 21.9, 2 x 4: 22.4, 4 x 4: 23.5 (64 / 56 / 72 / 128 registers)
#define AMX_TIED_BOUND_NBUF 2
#endif
typedef float tied_f2 __attribute__((ext_vector_type(2)));
typedef uint32_t tied_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ tied_u4 tied_load_u4(const unsigned short* tab, uint32_t row_off, uint32_t lane_off) {
    typedef const __attribute__((address_space(1))) char* gptr;
    gptr row = (gptr)tab + row_off;
    asm("" : "+s"(row));
    asm("" : "+v"(lane_off));
    return *(const __attribute__((address_space(1))) tied_u4*)(row + lane_off);
}

// Work to XCDs: a row has S = mix_pad / 512 (rounded up) segments.  The first 8 (S / 8) keep the fixed home XCD segment % 8 (each
// XCD's L2 then holds ITS segments of the rows the batch's frames share).  The remaining R = S % 8 would leave 8 - R XCDs idle a
// third of the time (config 3: S = 20, XCDs 0-3 three segments per frame, 4-7 two: measured as 21 us of ARITHMETIC alone, where a
// balanced chip needs 17), so their R x T (segment, frame) units are dealt round the XCDs: unit q = t R + j on XCD q % 8.
__global__ __launch_bounds__(64) void tied_bound8_kernel(const unsigned short* __restrict__ g_aup, const float* __restrict__ g_amax,
                                                        const uint2* __restrict__ g_near, int n_mix,
                                                        int mix_pad, int n_tiles, float* __restrict__ g_thr, float* __restrict__ g_thr_m,
                                                        int T, int fixed_per_xcd, int rest) {
    constexpr int B = AMX_TIED_BOUND_B, NBUF = AMX_TIED_BOUND_NBUF;  // rows per buffer, buffers: (NBUF - 1) B rows of 1 KB in flight per wave
    static_assert(kTiedNear % (NBUF * B) == 0 && B % 2 == 0, "whole rounds of buffers, row pairs");
    const int lane = threadIdx.x;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    int       t, seg;
    if (idx < fixed_per_xcd * T) {
        t   = idx / fixed_per_xcd;
        seg = xcd + 8 * (idx - t * fixed_per_xcd);
    }
    else {
        const int q = xcd + 8 * (idx - fixed_per_xcd * T);
        if (q >= rest * T)
            return;
        t   = q / rest;
        seg = 8 * fixed_per_xcd + (q - t * rest);
    }
    const int m = 8 * (seg * 64 + lane);  // mix_pad is a multiple of 64: m + 7 < mix_pad with m
    const bool      in = m < mix_pad;  // (the last segment's upper lanes: they read the row's first bytes and store nothing)
    const uint2*    nr = g_near + (size_t)t * kTiedNear;  // .x = list position, .y = distance bits (little-endian halves of the key)
    const uint32_t  row_bytes = (uint32_t)mix_pad * 2u, lane_off = in ? (uint32_t)m * 2u : 0u;
    tied_u4         buf[NBUF][B];
    float           u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        u[j] = FLT_MAX;
    auto fetch = [&](tied_u4 (&bf)[B], const uint2* k) {
#pragma unroll
        for (int i = 0; i < B; ++i)
            bf[i] = tied_load_u4(g_aup, k[i].x * row_bytes, lane_off);
    };
    auto sum = [&](const tied_u4 (&bf)[B], const uint2* d) {
#pragma unroll
        for (int i = 0; i < B; i += 2) {
            const float da = __uint_as_float(d[i].y), db = __uint_as_float(d[i + 1].y);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const tied_f2 a = tied_f2{__uint_as_float(bf[i][w] << 16), __uint_as_float(bf[i][w] & 0xffff0000u)} + tied_f2{da, da};
                const tied_f2 c = tied_f2{__uint_as_float(bf[i + 1][w] << 16), __uint_as_float(bf[i + 1][w] & 0xffff0000u)} + tied_f2{db, db};
                u[2 * w]        = fminf(fminf(u[2 * w], a.x), c.x);
                u[2 * w + 1]    = fminf(fminf(u[2 * w + 1], a.y), c.y);
            }
        }
    };
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j)
        fetch(buf[j], nr + j * B);
#pragma unroll 1
    for (int r = 0; r < kTiedNear; r += NBUF * B) {  // (a rolled loop: unrolled, the scheduler lifts all 64 loads to the top -- 394 registers)
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            const int nxt = r + (j + NBUF - 1) * B;  // the batch NBUF - 1 ahead goes into the buffer summed last
            if (nxt < kTiedNear)
                fetch(buf[(j + NBUF - 1) % NBUF], nr + nxt);
            sum(buf[j], nr + r + j * B);
        }
    }
    if (!in)
        return;  // whole groups of eight lanes (a tile) leave together
    // tau' = 2^-21 (2 max|a^| + |U|)
    float thr8[8], thr = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        thr8[j] = -__builtin_inff();
        if (m + j < n_mix) {
            thr8[j] = u[j] + (4.76837158e-7f * (2.f * g_amax[m + j] + fabsf(u[j])) + 1e-30f);
            if (!(thr8[j] == thr8[j]))
                thr8[j] = __builtin_inff();
        }
        thr = fmaxf(thr, thr8[j]);
    }
    float* out = g_thr_m + (size_t)t * mix_pad + m;  // the mixtures' own thresholds: tied_pruned_kernel screens with them (-inf: padding)
    *(float4*)out       = make_float4(thr8[0], thr8[1], thr8[2], thr8[3]);
    *(float4*)(out + 4) = make_float4(thr8[4], thr8[5], thr8[6], thr8[7]);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1)
        thr = fmaxf(thr, __shfl_xor(thr, o));
    const int tile = m >> 6;
    if ((lane & 7) == 0 && tile < n_tiles)
        g_thr[(size_t)t * n_tiles + tile] = thr;
}

// One workgroup of 16 waves per frame: ThrG = max over the tiles, then the densities with fl32(aminG[k] + dist) <= ThrG, ascending,
// with their distances and log-normalisation terms.  Wave w tests densities [256 w, 256 w + 256) of every 4096 (one trip to memory:
// the frame's whole list is a handful of trips deep, not K / 256), the waves' counts meet in LDS, and each wave appends behind the
// waves before it.  lk / ld / ll [T][Kpad], ln [T].
constexpr int kTiedListWaves = 16;

__global__ __launch_bounds__(64 * kTiedListWaves) void tied_list_kernel(const float* __restrict__ g_dt, const float* __restrict__ g_amin_all,
                                                                       const float* __restrict__ g_thr, const float* __restrict__ g_ln32, int K,
                                                                       int Kpad, int n_tiles, uint32_t* __restrict__ g_lk,
                                                                       float* __restrict__ g_ld, float* __restrict__ g_ll, int* __restrict__ g_ln,
                                                                       unsigned long long* __restrict__ g_examined, unsigned long long examined,
                                                                       unsigned long long* __restrict__ g_near) {
    constexpr int    NWV = kTiedListWaves;
    __shared__ float s_thr[NWV];
    __shared__ int   s_cnt[NWV];
    const int        t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the denominator of the survivor statistic travels with the numerator (the host reads both from ONE asynchronous copy: counting
    // the submitted triples on the host instead made the ratio wrong whenever the host ran ahead of the device)
    if (g_examined && t == 0 && threadIdx.x == 0)
        atomicAdd(g_examined, examined);
    // the bound kernel has read the frame's near keys: back to the empty state, for the atomic minima of the next call
    if (threadIdx.x < kTiedNear)
        g_near[(size_t)t * kTiedNear + threadIdx.x] = kTiedNearInit;
    float thr = -__builtin_inff();
    for (int j = threadIdx.x; j < n_tiles; j += 64 * NWV)
        thr = fmaxf(thr, g_thr[(size_t)t * n_tiles + j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        thr = fmaxf(thr, __shfl_xor(thr, o));
    if (lane == 0)
        s_thr[wave] = thr;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NWV; ++w)
        thr = fmaxf(thr, s_thr[w]);
    const float* row  = g_dt + (size_t)t * Kpad;
    uint32_t*    lk   = g_lk + (size_t)t * Kpad;
    float*       ld   = g_ld + (size_t)t * Kpad;
    float*       ll   = g_ll + (size_t)t * Kpad;
    int          base = 0;
    for (int r0 = 0; r0 < Kpad; r0 += 256 * NWV) {
        const int          kb = r0 + 256 * wave;
        float              dv[4], am[4], lv[4];
        unsigned long long mask[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kb + 64 * u + lane;
            dv[u]       = k < Kpad ? row[k] : __builtin_inff();
            am[u]       = k < Kpad ? g_amin_all[k] : __builtin_inff();
            lv[u]       = k < K ? g_ln32[k] : 0.f;
        }
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kb + 64 * u + lane;
            mask[u]     = __ballot(k < K && (am[u] + dv[u]) <= thr);  // k < K: +inf <= thr when the threshold is +inf itself
            cnt += __popcll(mask[u]);
        }
        if (lane == 0)
            s_cnt[wave] = cnt;
        __syncthreads();
        int pos = base;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            const int c = s_cnt[w];
            pos += w < wave ? c : 0;
            base += c;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if ((mask[u] >> lane) & 1ull) {
                const int p = pos + __popcll(mask[u] & ((1ull << lane) - 1ull));
                lk[p]       = (uint32_t)(kb + 64 * u + lane);
                ld[p]       = dv[u];
                ll[p]       = lv[u];  // logNorm (an f32 value) travels with the entry
            }
            pos += __popcll(mask[u]);
        }
    }
    if (threadIdx.x == 0)
        g_ln[t] = base;
}

// tab[(row_off + lane_off) bytes] as the (scalar base, 32-bit lane offset) form of global_load: row_off is wave-uniform, so the row
// address is scalar arithmetic and no 64-bit address per lane exists.  The two empty asm statements keep the compiler from
// re-associating the sum into (tab + lane_off) + row_off, which it otherwise hoists and pays for with a 64-bit vector add per load.
__device__ __forceinline__ float tied_load(const float* tab, uint32_t row_off, uint32_t lane_off) {
    typedef const __attribute__((address_space(1))) char* gptr;
    gptr row = (gptr)tab + row_off;
    asm("" : "+s"(row));
    asm("" : "+v"(lane_off));
    return *(const __attribute__((address_space(1))) float*)(row + lane_off);
}

struct TiedMax {  // MaxState of gmm.hip (the reference's rule), restated here to keep this file self-contained
    float    best   = FLT_MAX;
    double   best_d = (double)FLT_MAX;
    uint32_t idx    = 0xffffffffu;
    __device__ __forceinline__ void add(double c64, float dist, uint32_t k) {
        const double s = c64 + (double)dist;
        if (best_d > s) {
            best   = (float)s;
            best_d = (double)best;
            idx    = k;
        }
    }
};

constexpr int kTiedSeg  = 256;  // survivors the unscreened path buffers per wave; a fuller list is worked off and the scan resumes
constexpr int kTiedCap  = 256;  // entries of a (frame, tile) survivor list in LDS; a longer one takes the unscreened path

// Per (frame, 64 list entries, tile): the 64-bit mask of the entries that pass the TILE's test.  lane = tile, so the test reads the
// transposed table amin_t[k][tile] -- one coalesced row per list entry for 64 tiles -- where a wave of tied_pruned_kernel would gather
// 4 bytes per entry for ONE tile (a scattered 64-lane gather costs the CU 150 cycles, tools/gather_probe.hip).  A wave tests 16
// entries (one trip to memory: the kernel has 3 x T workgroups, its time is the depth of a wave's chain of loads) and stores its 16
// bits of the mask; the entries sit in lane registers, the entry under test is a v_readlane scalar.
// masks [T][Kpad / 64][tiles_pad] of 64 bits = 4 x 16.
constexpr int kTiedMaskWaves = 16;

__global__ __launch_bounds__(64 * kTiedMaskWaves) void tied_mask_kernel(const uint32_t* __restrict__ g_lk, const float* __restrict__ g_ld,
                                                                       const int* __restrict__ g_ln, const float* __restrict__ g_amin_t,
                                                                       const float* __restrict__ g_thr, int Kpad, int n_tiles, int tiles_pad,
                                                                       unsigned short* __restrict__ g_mask) {
    const int       lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int       t = blockIdx.y, tile = blockIdx.x * 64 + lane;
    const bool      live = tile < n_tiles;
    const float     Thr  = live ? g_thr[(size_t)t * n_tiles + tile] : -__builtin_inff();
    const int       nl   = g_ln[t];
    const uint32_t* lk   = g_lk + (size_t)t * Kpad;
    const float*    ld   = g_ld + (size_t)t * Kpad;
    const uint32_t  row_bytes = (uint32_t)tiles_pad * 4u, lane_off = (uint32_t)tile * 4u;
    const int       n_groups = 4 * ((nl + 63) / 64);  // whole 64-entry chunks: tied_pruned_kernel reads the masks chunk by chunk
    for (int g = wave; g < n_groups; g += kTiedMaskWaves) {
        const int      e  = 16 * g + (lane & 15);
        const uint32_t rk = e < nl ? lk[e] : 0u;
        const float    rd = e < nl ? ld[e] : __builtin_nanf("");  // NaN: never passes
        float          a[16];
        uint32_t       mask = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            a[u] = tied_load(g_amin_t, (uint32_t)__builtin_amdgcn_readlane((int)rk, u) * row_bytes, lane_off);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rd), u));
            mask |= live && (a[u] + d) <= Thr ? 1u << u : 0u;
        }
        g_mask[(((size_t)t * (Kpad / 64) + (g >> 2)) * tiles_pad + tile) * 4 + (g & 3)] = (unsigned short)mask;
    }
}

// The unscreened path of tied_pruned_kernel: the tile's test over the frame's list and the reference's f64 rule over EVERY survivor
// for all 64 mixtures.  Phase 1 (lane = list entry) compacts the survivors -- position, distance, log-normalisation term -- into
// LDS; phase 2 (lane = mixture) runs the rule over them in ascending order with PF rows of the weight table in flight.
struct TiedLds {
    uint32_t k[1][kTiedSeg];
    float    d[1][kTiedSeg];
    double   l[1][kTiedSeg];
};
struct TiedList {  // the (frame, tile) survivor list of tied_pruned_kernel: row offset in the weight table, distance, logNorm, density
    uint32_t rk[kTiedCap], d[kTiedCap], l[kTiedCap], k[kTiedCap];
};

__device__ __forceinline__ void tied_unscreened(TiedLds& lds, int wave, int lane, const uint32_t* __restrict__ lk, const float* __restrict__ ld,
                                                int nl, const float* __restrict__ arow, float Thr, const float* __restrict__ g_m2lw_t,
                                                const double* __restrict__ g_ln64, int mix_pad, int m, TiedMax& st) {
    int ib = 0;
    while (ib < nl) {
        // ---- phase 1
        int n = 0;
        for (; ib < nl && n + 128 <= kTiedSeg; ib += 128) {  // two 64-entry steps per trip: list loads, then gathers, then the tests
            uint32_t k[2];
            float    dv[2], am[2];
            double   ln[2];
            bool     in[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = ib + 64 * u + lane;
                in[u]       = i < nl;
                k[u]        = in[u] ? lk[i] : 0u;
                dv[u]       = in[u] ? ld[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                am[u] = arow[k[u]];
                ln[u] = g_ln64[k[u]];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool               rel  = in[u] && (am[u] + dv[u]) <= Thr;
                const unsigned long long mask = __ballot(rel);
                if (rel) {
                    const int pos    = n + __popcll(mask & ((1ull << lane) - 1ull));
                    lds.k[wave][pos] = k[u];
                    lds.d[wave][pos] = dv[u];
                    lds.l[wave][pos] = ln[u];
                }
                n += __popcll(mask);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 2
        constexpr int PF = 8;
        for (int i = 0; i < n; i += PF) {
            float w[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int ii = i + j < n ? i + j : n - 1;
                w[j]         = g_m2lw_t[(size_t)lds.k[wave][ii] * mix_pad + m];
            }
#pragma unroll
            for (int j = 0; j < PF; ++j)
                if (i + j < n)
                    st.add((double)w[j] + lds.l[wave][i + j], lds.d[wave][i + j], lds.k[wave][i + j]);
        }
        __builtin_amdgcn_wave_barrier();  // the lists are rewritten by the next segment
    }
}

// tab[byte_off] with a scalar table address and a 32-bit byte offset per lane
__device__ __forceinline__ float tied_load_v(const float* tab, uint32_t byte_off) {
    typedef const __attribute__((address_space(1))) char* gptr;
    gptr base = (gptr)tab;
    asm("" : "+s"(base));
    asm("" : "+v"(byte_off));
    return *(const __attribute__((address_space(1))) float*)(base + byte_off);
}

// One wave (= one workgroup: no wave waits for its neighbours' lists) per (64-mixture tile, frame).
// Phase 1 (lane = list entry): the tile's masks (tied_mask_kernel) compact the frame's list into LDS -- weight-table row offset,
// distance, logNorm, density -- in list order.
// Phase 2 (lane = mixture): a survivor of the TILE's test is a candidate for one or two of the tile's 64 mixtures, so the lane first
// screens its OWN mixture in f32 with the threshold the bound kernel derived for it: s^ = fl32(a^[k][m] + dist) <= U[t][m] + tau'
// holds for every candidate of mixture m (file header).  The list positions that pass -- one bit per position and lane -- are again
// a subsequence with every candidate in it, and only those go through the reference's f64 rule, in list order (phase 3).
//
// The kernel is bound by the NUMBER of instructions a wave issues (a wave issues one per 4 cycles whatever the unit; with 40 waves
// per SIMD the row gather itself is a quarter of the time), so the loop is built to need few:
//   * 16 consecutive entries of a list field sit in one register (entry j in lane j of every row of 16 lanes); the entry under test
//     reaches all lanes as the DPP operand row_newbcast:j of the add that consumes it -- no instruction of its own, where a scalar
//     round trip costs v_readlane + s_add + s_addc per row -- and the hit is shifted into the mask by ONE v_addc (mask + mask +
//     carry): 6 instructions per survivor (offset, load, two sums, compare, shift-in), where position lists and scalar row
//     addresses took 18.
//   * a^ = fl32(weight + logNorm) is what the a^ table holds, bit for bit; the loop reads the WEIGHT table and adds, so the kernel
//     stays on one 164 MB table and the candidates' weights (fetched again in phase 3, one 64-byte sector per lane and candidate)
//     are L2 hits.
// A list longer than kTiedCap or a table beyond 32-bit offsets takes tied_unscreened.
__global__ __launch_bounds__(64) void tied_pruned_kernel(const unsigned long long* __restrict__ g_mask, const uint32_t* __restrict__ g_lk,
                                                        const float* __restrict__ g_ld, const float* __restrict__ g_ll,
                                                        const int* __restrict__ g_ln, const float* __restrict__ g_amin,
                                                        const float* __restrict__ g_thr, const float* __restrict__ g_m2lw_t,
                                                        const float* __restrict__ g_thr_m, const double* __restrict__ g_ln64, int Kpad, int T,
                                                        int n_mix, int mix_pad, int n_tiles, int tiles_pad, float* __restrict__ g_scores,
                                                        uint32_t* __restrict__ g_best, unsigned long long* __restrict__ g_survivors) {
    __shared__ union {
        TiedLds  u;
        TiedList s;
    } lds;
    const int lane = threadIdx.x;
    // Workgroup b (in launch order: x fastest) runs on XCD b % 8, and each XCD has its own L2.  A row of the weight table is wanted
    // by ~3 of a 256-frame batch's frames, so all frames of one tile go to ONE XCD, back to back: the grid is (8 T, tiles / 8),
    // xcd = x % 8, frame = x / 8, tile = 8 y + xcd.  The tile's 1 MB slice of the table then comes from HBM once instead of once per
    // frame that wants it.
    // The tiles left over after the whole rows of eight (config 3: 157 = 19 x 8 + 5) would keep 8 - R XCDs idle for the last row: their
    // R x T (tile, frame) items are dealt round the XCDs instead, item q = t R + j on XCD q % 8 (as in tied_bound8_kernel).
    int tile = blockIdx.y * 8 + (blockIdx.x & 7), t = blockIdx.x >> 3;
    const int rest = n_tiles & 7;
    if (rest != 0 && (int)blockIdx.y == (n_tiles >> 3)) {
        const int q = (blockIdx.x & 7) + 8 * (blockIdx.x >> 3);
        if (q >= rest * T)
            return;
        t    = q / rest;
        tile = (n_tiles & ~7) + (q - t * rest);
    }
    const int       m      = tile * 64 + lane;
    const float     thr_m  = g_thr_m[(size_t)t * mix_pad + m];
    const bool      narrow = (unsigned long long)Kpad * mix_pad * 4ull < (1ull << 32);  // table offsets fit 32 bits
    const uint32_t  m_off  = (uint32_t)m * 4u, row_bytes = (uint32_t)mix_pad * 4u;
    const uint32_t* lk     = g_lk + (size_t)t * Kpad;
    const float*    ld     = g_ld + (size_t)t * Kpad;
    const float*    ll     = g_ll + (size_t)t * Kpad;
    constexpr int   NW     = kTiedCap / 32;  // 32-entry words of a list
    // ---- phase 1.  The first 256 list entries and their masks are fetched before the list's length is known (the rows are Kpad
    // long; what lies past the end is masked out below): one trip to memory for the usual list, not one per 64 entries.
    int  n       = 0;
    auto compact = [&](unsigned long long mask, uint32_t k, float d, float l) {
        const int pos = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (((mask >> lane) & 1ull) && pos < kTiedCap) {
            lds.s.rk[pos] = k * row_bytes;
            lds.s.d[pos]  = __float_as_uint(d);
            lds.s.l[pos]  = __float_as_uint(l);
            lds.s.k[pos]  = k;
        }
        n += __popcll(mask);
    };
    constexpr int      NC = 4;
    unsigned long long mk[NC];
    uint32_t           kk[NC];
    float              dd[NC], l4[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const bool ok = 64 * c < Kpad;
        const int  e  = ok ? 64 * c + lane : 0;
        mk[c]         = ok ? g_mask[((size_t)t * (Kpad / 64) + c) * tiles_pad + tile] : 0ull;  // wave-uniform
        kk[c]         = lk[e];
        dd[c]         = ld[e];
        l4[c]         = ll[e];
    }
    const int nl = g_ln[t];
#pragma unroll
    for (int c = 0; c < NC; ++c)
        compact(64 * c < nl ? mk[c] : 0ull, kk[c], dd[c], l4[c]);  // (the mask kernel writes the chunks that hold list entries only)
    for (int c = NC; 64 * c < nl; ++c) {
        const int e = 64 * c + lane;  // < Kpad; the mask is clear past the list's end
        compact(g_mask[((size_t)t * (Kpad / 64) + c) * tiles_pad + tile], lk[e], ld[e], ll[e]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    TiedMax st;
    if (narrow && n <= kTiedCap) {
        // ---- phase 2
        uint32_t hit[NW];  // bit 31 - j of hit[w]: list position 32 w + j
#pragma unroll
        for (int w = 0; w < NW; ++w)
            hit[w] = 0u;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (32 * w >= n)
                break;
            // two blocks of 16 entries; entry j of a block sits in lane j of EVERY row of 16 lanes, so that the DPP operand
            // row_newbcast:j hands it to all lanes inside the add that consumes it
            const int  e0   = 32 * w + (lane & 15), e1 = e0 + 16;
            const bool two  = 32 * w + 16 < n;  // the second block exists
            const int  rk0  = e0 < n ? (int)lds.s.rk[e0] : 0;          // row 0 stands in past the end ...
            const int  d0   = e0 < n ? (int)lds.s.d[e0] : 0x7fc00000;  // ... with a NaN distance: never a candidate
            const int  l0   = e0 < n ? (int)lds.s.l[e0] : 0;
            const int  rk1  = e1 < n ? (int)lds.s.rk[e1] : 0;
            const int  d1   = e1 < n ? (int)lds.s.d[e1] : 0x7fc00000;
            const int  l1   = e1 < n ? (int)lds.s.l[e1] : 0;
            float      a[32];
            uint32_t   hm = 0u;
#define AMX_TIED_BC(V, J) __builtin_amdgcn_update_dpp(0, V, 0x150 + (J), 0xf, 0xf, false) /* row_newbcast:J */
#define AMX_TIED_ROW0(J) a[J] = tied_load_v(g_m2lw_t, (uint32_t)AMX_TIED_BC(rk0, J) + m_off);
#define AMX_TIED_ROW1(J) a[16 + J] = tied_load_v(g_m2lw_t, (uint32_t)AMX_TIED_BC(rk1, J) + m_off);
#define AMX_TIED_TEST(A, L, D, J)                                                                                             \
    {                                                                                                                         \
        const float s_ = (A + __int_as_float(AMX_TIED_BC(L, J))) + __int_as_float(AMX_TIED_BC(D, J));                         \
        asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hm) : "v"(s_), "v"(thr_m) : "vcc");      \
    }
#define AMX_TIED_TEST0(J) AMX_TIED_TEST(a[J], l0, d0, J)
#define AMX_TIED_TEST1(J) AMX_TIED_TEST(a[16 + J], l1, d1, J)
#define AMX_TIED_16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
            AMX_TIED_16(AMX_TIED_ROW0)
            if (two) {
                AMX_TIED_16(AMX_TIED_ROW1)
            }
            AMX_TIED_16(AMX_TIED_TEST0)
            if (two) {
                AMX_TIED_16(AMX_TIED_TEST1)
            }
            else
                hm <<= 16;
#undef AMX_TIED_16
#undef AMX_TIED_TEST1
#undef AMX_TIED_TEST0
#undef AMX_TIED_TEST
#undef AMX_TIED_ROW1
#undef AMX_TIED_ROW0
#undef AMX_TIED_BC
            hit[w] = hm;
        }
#if AMX_TIED_EXP & 4
        {
            int cnt = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                cnt += __popc(hit[w]);
            if (m < n_mix)
                atomicAdd(&g_tied_hist[min(cnt, 31)], 1ull);
            if (lane == 0)
                atomicAdd(&g_tied_hist[32 + min(n / 8, 31)], 1ull);
        }
#endif
        // ---- phase 3: 64 list positions at a time, every pass takes each lane's first remaining one (entry from LDS, the lane's own
        // weight from the table) through the reference's rule -- list order within a lane is all the rule needs.  One or two passes
        // per 64 positions for most waves; a lane whose threshold is infinite walks its whole list.
#pragma unroll
        for (int w = 0; w < NW; w += 2) {
            if (32 * w >= n)
                break;
            unsigned long long h64 = ((unsigned long long)hit[w] << 32) | hit[w + 1];
            while (__any(h64 != 0ull)) {
                int   pos[4];
                float wt[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // up to four candidates' weights in flight: one trip, not four
                    pos[j] = -1;
                    wt[j]  = 0.f;
                    if (h64 != 0ull) {
                        const int b = __builtin_clzll(h64);
                        h64 &= ~(0x8000000000000000ull >> b);
                        pos[j] = 32 * w + b;
                        wt[j]  = tied_load_v(g_m2lw_t, lds.s.rk[pos[j]] + m_off);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (pos[j] >= 0)
                        st.add((double)wt[j] + (double)__uint_as_float(lds.s.l[pos[j]]), __uint_as_float(lds.s.d[pos[j]]), lds.s.k[pos[j]]);
            }
        }
    }
    else {
        __builtin_amdgcn_wave_barrier();
        tied_unscreened(lds.u, 0, lane, lk, ld, nl, g_amin + (size_t)tile * Kpad, g_thr[(size_t)t * n_tiles + tile], g_m2lw_t, g_ln64, mix_pad,
                        m, st);
    }
    if (m < n_mix) {
        g_scores[(size_t)t * n_mix + m] = 0.5f * st.best;
        if (g_best)
            g_best[(size_t)t * n_mix + m] = st.idx;
    }
    // statistics for the host's dense / pruned decision: spread over kTiedCounters addresses (40 000 atomics on ONE address cost
    // 0.37 ms, more than the rest of this kernel)
    if (lane == 0 && g_survivors)
        atomicAdd(g_survivors + ((tile * 7 + t) & (kTiedCounters - 1)), (unsigned long long)n);
}

}  // namespace amx

// amin[tile][k] = min over the real mixtures of the tile of a^[k][m]; aminG[k] = min over the tiles.  Returns one device table
// [(n_tiles + 1)][Kpad] (+inf padded), row n_tiles = aminG, followed by amin transposed, [Kpad][tiles_pad].
// bf16 that is >= the f32 value (NaN / inf pass through; +0 for the padding columns)
static unsigned short tied_bf16_up(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7f800000u) == 0x7f800000u || (u & 0xffffu) == 0)
        return (unsigned short)(u >> 16);
    // positive: truncation rounds down -> next bf16 up; negative: truncation (towards zero) already rounds up
    return (unsigned short)((u >> 16) + ((u >> 31) ? 0u : 1u));
}

extern "C" int amx_internal_gmm_tied_create(int K, int n_mix, int mix_pad, const float* ahat_t_host, float** d_amin, unsigned short** d_aup) {
    const int          n_tiles = mix_pad / 64, Kpad = (K + 63) & ~63;
    {
        std::vector<unsigned short> up((size_t)K * mix_pad);
        for (size_t i = 0; i < up.size(); ++i)
            up[i] = tied_bf16_up(ahat_t_host[i]);
        *d_aup = nullptr;
        AMX_HIP(hipMalloc((void**)d_aup, up.size() * 2));
        AMX_HIP(hipMemcpy(*d_aup, up.data(), up.size() * 2, hipMemcpyHostToDevice));
    }
    std::vector<float> amin((size_t)(n_tiles + 1) * Kpad, __builtin_inff());
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < n_mix; ++i) {
            const float a = ahat_t_host[(size_t)k * mix_pad + i];
            float&      v = amin[(size_t)(i >> 6) * Kpad + k];
            v             = a < v ? a : v;
            float& g      = amin[(size_t)n_tiles * Kpad + k];
            g             = a < g ? a : g;
        }
    // the same numbers transposed, [Kpad][tiles_pad] (+inf padded), behind them: tied_tile_list_kernel tests 64 tiles per load
    const int    tiles_pad = (n_tiles + 63) & ~63;
    const size_t t0        = amin.size();
    amin.resize(t0 + (size_t)Kpad * tiles_pad, __builtin_inff());
    for (int j = 0; j < n_tiles; ++j)
        for (int k = 0; k < K; ++k)
            amin[t0 + (size_t)k * tiles_pad + j] = amin[(size_t)j * Kpad + k];
    *d_amin = nullptr;
    AMX_HIP(hipMalloc((void**)d_amin, amin.size() * sizeof(float)));
    AMX_HIP(hipMemcpy(*d_amin, amin.data(), amin.size() * sizeof(float), hipMemcpyHostToDevice));
    return AMX_OK;
}

namespace {
constexpr int kTiedFrames = 4096;  // frames per pass of amx_internal_gmm_tied_score: bounds the workspace
struct TiedWs {
    float *             dt, *ld, *ll, *thr, *thr_m;
    uint32_t*           lk;
    unsigned long long* near;  // [kTiedFrames][kTiedNear] keys, FIRST and of fixed size: its place and its empty state survive calls of any shape
    int*                ln;
    unsigned long long* mask;
    size_t              bytes;
};
TiedWs tied_ws(void* base, int K, int T, int mix_pad) {
    const size_t Kpad = (size_t)((K + 63) & ~63), n_tiles = (size_t)mix_pad / 64, tiles_pad = (n_tiles + 63) & ~(size_t)63;
    auto         al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    char*        p  = (char*)base;
    TiedWs       w;
    T      = std::min(T, kTiedFrames);
    w.near = (unsigned long long*)p;
    p += al((size_t)kTiedFrames * amx::kTiedNear * 8);
    w.dt = (float*)p;
    p += al((size_t)T * Kpad * 4);
    w.lk = (uint32_t*)p;
    p += al((size_t)T * Kpad * 4);
    w.ld = (float*)p;
    p += al((size_t)T * Kpad * 4);
    w.ll = (float*)p;
    p += al((size_t)T * Kpad * 4);
    w.ln = (int*)p;
    p += al((size_t)T * 4);
    w.thr = (float*)p;
    p += al((size_t)T * n_tiles * 4);
    w.thr_m = (float*)p;
    p += al((size_t)T * mix_pad * 4);
    w.mask = (unsigned long long*)p;
    p += al((size_t)T * (Kpad / 64) * tiles_pad * 8);
    w.bytes = (size_t)(p - (char*)base);
    return w;
}
}  // namespace

extern "C" size_t amx_internal_gmm_tied_workspace(int K, int T, int mix_pad) {
    return tied_ws(nullptr, K, T, mix_pad).bytes;
}

// where gmm_dist_kernel may write the frame-major distances itself (then amx_internal_gmm_tied_score skips its transposing kernel):
// the call must be ONE pass, and the model's list must name every density at most once
extern "C" float* amx_internal_gmm_tied_dt(void* workspace, int K, int T, int have_positions) {
    return have_positions && T <= kTiedFrames ? tied_ws(workspace, K, T, 64).dt : nullptr;
}

// the near keys of a workspace (gmm_dist_list_kernel keeps the minima itself when the build has 64 classes = its lanes), and their
// empty state, to be set once per allocation and after a failed call
extern "C" unsigned long long* amx_internal_gmm_tied_near(void* workspace) {
    return amx::kTiedNear == 64 ? (unsigned long long*)workspace : nullptr;
}
extern "C" int amx_internal_gmm_tied_near_init(amx_ctx* ctx, void* workspace) {
    const int n = kTiedFrames * amx::kTiedNear;
    hipLaunchKernelGGL(amx::tied_near_init_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (unsigned long long*)workspace, n);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

extern "C" int amx_internal_gmm_tied_score(amx_ctx* ctx, const float* dist_dev, const uint32_t* k_dens_dev, int K, int T, int Tpad, int n_mix,
                                           int mix_pad, const unsigned short* aup, const float* amax, const float* m2lw_t, const float* ahat_t,
                                           const double* ln64, const float* ln32, const float* amin, void* workspace, float* scores,
                                           uint32_t* best, unsigned long long* survivors_dev, int dt_written, int near_written) {
    if (T <= 0)
        return AMX_OK;
    const int    Kpad = (K + 63) & ~63, n_tiles = mix_pad / 64, tiles_pad = (n_tiles + 63) & ~63;
    const TiedWs w    = tied_ws(workspace, K, T, mix_pad);
    const float* amin_t = amin + (size_t)(n_tiles + 1) * Kpad;
    for (int t0 = 0; t0 < T; t0 += kTiedFrames) {  // dist is [n_dens][Tpad]: a pass is a column range
        const int Tc = std::min(kTiedFrames, T - t0);
        float*    sc = scores + (size_t)t0 * n_mix;
        uint32_t* bd = best ? best + (size_t)t0 * n_mix : nullptr;
        if (!dt_written)  // (gmm_dist_kernel wrote w.dt: amx_internal_gmm_tied_dt)
            hipLaunchKernelGGL(amx::tied_transpose_kernel, dim3(Kpad / 64, (Tc + 63) / 64), dim3(256), 0, ctx->stream, dist_dev + t0, k_dens_dev,
                               K, Kpad, Tc, Tpad - t0, Tpad, w.dt);
        if (!(dt_written && near_written))
            hipLaunchKernelGGL(amx::tied_near_kernel, dim3(Tc), dim3(amx::kTiedNearThreads), 0, ctx->stream, w.dt, K, Kpad, w.near);
        const int bound_spx = ((mix_pad / 2 + 255) / 256 + 7) / 8;   // 1 KB segments of a table row per XCD
        if ((unsigned long long)K * (unsigned long long)mix_pad * 2ull < (1ull << 32)) {
            const int segs = (mix_pad + 511) / 512, fixed = segs / 8, rest = segs % 8;
            hipLaunchKernelGGL(amx::tied_bound8_kernel, dim3(8 * (fixed * Tc + (rest * Tc + 7) / 8)), dim3(64), 0, ctx->stream, aup, amax,
                               (const uint2*)w.near, n_mix, mix_pad, n_tiles, w.thr, w.thr_m, Tc, fixed, rest);
        }
        else
            hipLaunchKernelGGL(amx::tied_bound_kernel, dim3(8 * bound_spx * Tc), dim3(256), 0, ctx->stream, aup, amax, (const uint2*)w.near, n_mix, mix_pad,
                               n_tiles, w.thr, w.thr_m, bound_spx);
        hipLaunchKernelGGL(amx::tied_list_kernel, dim3(Tc), dim3(64 * amx::kTiedListWaves), 0, ctx->stream, w.dt, amin + (size_t)n_tiles * Kpad, w.thr, ln32, K, Kpad,
                           n_tiles, w.lk, w.ld, w.ll, w.ln, survivors_dev ? survivors_dev + amx::kTiedCounters : nullptr,
                           (unsigned long long)K * (unsigned long long)Tc * (unsigned long long)n_tiles, w.near);
        hipLaunchKernelGGL(amx::tied_mask_kernel, dim3(tiles_pad / 64, Tc), dim3(64 * amx::kTiedMaskWaves), 0, ctx->stream, w.lk, w.ld, w.ln,
                           amin_t, w.thr, Kpad, n_tiles, tiles_pad, (unsigned short*)w.mask);
        hipLaunchKernelGGL(amx::tied_pruned_kernel, dim3(8 * Tc, (n_tiles + 7) / 8), dim3(64), 0, ctx->stream, w.mask, w.lk, w.ld, w.ll, w.ln,
                           amin, w.thr, m2lw_t, w.thr_m, ln64, Kpad, Tc, n_mix, mix_pad, n_tiles, tiles_pad, sc, bd, survivors_dev);
    }
    AMX_HIP(hipGetLastError());
#if AMX_TIED_EXP & 4
    static int calls = 0;
    if (++calls == 20) {
        unsigned long long h[64];
        hipStreamSynchronize(ctx->stream);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(amx::g_tied_hist), sizeof h);
        fprintf(stderr, "tied candidates per lane:");
        for (int i = 0; i < 32; ++i)
            fprintf(stderr, " %llu", h[i]);
        fprintf(stderr, "\ntied survivors/8 per wave:");
        for (int i = 32; i < 64; ++i)
            fprintf(stderr, " %llu", h[i]);
        fprintf(stderr, "\n");
    }
#endif
    return AMX_OK;
}
