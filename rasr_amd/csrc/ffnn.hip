// ffnn.hip -- feed-forward NN emission scorer (MFMA GEMM chain) for gfx950 and the amx_ffnn_* ABI.
//
// Replaces Nn::BatchFeatureScorer::getScore's network_.forward(buffer_)
// (Nn/BatchFeatureScorer.cc:148-171 -> Nn/NeuralNetwork.cc:313-331,409-425): per layer
// cblas_sgemm (Nn/LinearLayer.cc:306-308) + addToAllColumns (:317-318) + activation
// (Nn/ActivationLayer.cc:272-282), output layer without softmax and with the scaled log-prior
// removed from the bias (Nn/LinearAndActivationLayer.hh:137-160); score = -activation.
//
// Mapping to the hardware
//   * every layer is one GEMM D[n][t] = sum_k W[n][k] * X[t][k]; both operands are K-contiguous
//     (W is [out x in] row-major, X is [frames x in] row-major), which is exactly the MFMA
//     operand shape: the weight matrix is the MFMA "A" operand, the frames are "B", so one lane
//     ends up holding 4 consecutive output units of ONE frame -> 8-byte (bf16) / 16-byte (f32)
//     row-major stores, and bias/activation/prior/negation are fused into that epilogue.
//   * bf16 path: 128(n) x 128(t) x 64(k) workgroup tile, 4 wavefronts (2x2), each 64x64 as 2x2
//     v_mfma_f32_32x32x16_bf16 tiles; operands staged HBM -> LDS with 16-byte
//     global_load_lds (no VGPR round trip) into two LDS buffers; the next K-tile's loads are in
//     flight while the current one is multiplied (raw s_barrier + counted vmcnt, never a full
//     drain inside the loop).  LDS rows are 128 B; 16-byte chunks are XOR-swizzled
//     (chunk ^ ((row>>1)&7)) on the global source address so that ds_read_b128 of a 32-row
//     MFMA fragment is bank-conflict free.
//   * fp32 path (parity mode): v_mfma_f32_32x32x2_f32, exact f32 -- a k-ordered fmaf chain -- so
//     scores are comparable with the CPU sgemm path at 1e-6.
//   * activations stay in HBM as bf16 between layers (288 GB: no need to fuse layers).
#include "common.hpp"

#include <hip/hip_bf16.h>

#include <algorithm>
#include <map>
#include <tuple>
#include <type_traits>
#include <cmath>
#include <cstring>

namespace amx {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short                              bf16_t;

typedef float  f32x2 __attribute__((ext_vector_type(2)));
typedef float  f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two f32 -> packed bf16 (round to nearest even); lowers to one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ bf16_t f2bf(float f) {
    return (bf16_t)(pack_bf16(f, 0.f) & 0xffffu);
}

static inline bf16_t f2bf_host(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u)
        return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template<int ACT>
__device__ __forceinline__ float activate(float v) {
    if (ACT == AMX_ACT_RELU)
        return v < 0.f ? 0.f : v;  // ensureMinimalValue(0)
    if (ACT == AMX_ACT_SIGMOID)
        return 1.f / (1.f + expf(-v));  // Math/FastMatrix.hh:802-808
    if (ACT == AMX_ACT_TANH)
        return tanhf(v);
    return v;
}

// ---------------------------------------------------------------------------------------------
// f32 frames -> bf16 [Tpad x Kpad], zero padded
__global__ __launch_bounds__(256) void pack_input_bf16(const float* __restrict__ x, int ldx, int T, int K, bf16_t* __restrict__ out,
                                                      int Kpad, int Tpad) {
    const long long n = (long long)Tpad * Kpad;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int   t = (int)(i / Kpad), k = (int)(i - (long long)t * Kpad);
        float v = (t < T && k < K) ? x[(size_t)t * ldx + k] : 0.f;
        out[i]  = f2bf(v);
    }
}

// ---- split-bf16 ("bf16x3") mode: v = hi + lo with hi = bf16(v), lo = bf16(v - hi); a product a b is taken as
// a_hi b_hi + a_lo b_hi + a_hi b_lo (the a_lo b_lo term is below 2^-17 |a b|), f32 accumulation -- three bf16 MFMA products
// instead of one, relative error of a product ~2^-16 instead of 2^-8.  Operand rows are [hi plane | lo plane]; the GEMM kernels
// stage both planes of a K-tile and issue the three products from one set of fragment reads (GemmCfg::X3).
// features -> [Tpad x 2 seg] bf16
__global__ __launch_bounds__(256) void pack_input_bf16x3(const float* __restrict__ x, int ldx, int T, int K, bf16_t* __restrict__ out,
                                                        int seg, int Tpad) {
    const long long n = (long long)Tpad * seg;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int    t = (int)(i / seg), k = (int)(i - (long long)t * seg);
        const float  v = (t < T && k < K) ? x[(size_t)t * ldx + k] : 0.f;
        const bf16_t hi = f2bf(v);
        bf16_t*      o = out + (size_t)t * 2 * seg + k;
        o[0]           = hi;
        o[seg]         = f2bf(v - __uint_as_float((unsigned)hi << 16));
    }
}

// last hidden activation -> f32 [T x H] (Nn::OnDemandFeatureScorer::forwardHiddenLayers keeps it per frame).
// MODE 0: f32 rows, 1: bf16 rows, 2: split bf16 rows [hi | lo], lo plane at column seg (value = hi + lo)
template<int MODE>
__global__ __launch_bounds__(256) void export_hidden_kernel(const void* __restrict__ src, int ld, int seg, int T, int H, float* __restrict__ out) {
    const long long n = (long long)T * H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int t = (int)(i / H), k = (int)(i - (long long)t * H);
        float     v;
        if (MODE == 0)
            v = ((const float*)src)[(size_t)t * ld + k];
        else if (MODE == 1)
            v = __uint_as_float((unsigned)((const bf16_t*)src)[(size_t)t * ld + k] << 16);
        else {
            const bf16_t* r = (const bf16_t*)src + (size_t)t * ld;
            v               = __uint_as_float((unsigned)r[k] << 16) + __uint_as_float((unsigned)r[seg + k] << 16);
        }
        out[i] = v;
    }
}

// Nn::LinearAndSoftmaxLayer::getScore (Nn/LinearAndActivationLayer.cc:154-160) for a list of (frame, emission) pairs:
// score = -bias[e] - W[e] . act[frame]  (bias has -alpha * log prior folded in; a disregarded class has a zero row and bias
// -FLT_MAX).  One wavefront per pair, lane-strided f32 partial sums + butterfly (the reference's sdot order is unspecified).
__global__ __launch_bounds__(256) void on_demand_kernel(const float* __restrict__ act, int H, const float* __restrict__ W, const float* __restrict__ bias,
                                                       const uint32_t* __restrict__ frame, const uint32_t* __restrict__ emission, int n_pairs,
                                                       float* __restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const int p    = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_pairs)
        return;
    const float* a = act + (size_t)frame[p] * H;
    const float* w = W + (size_t)emission[p] * H;
    float        s = 0.f;
    if ((H & 3) == 0) {
        for (int k = lane * 4; k < H; k += 256) {
            const float4 x = *(const float4*)(a + k), y = *(const float4*)(w + k);
            s              = fmaf(x.x, y.x, s);
            s              = fmaf(x.y, y.y, s);
            s              = fmaf(x.z, y.z, s);
            s              = fmaf(x.w, y.w, s);
        }
    }
    else
        for (int k = lane; k < H; k += 64)
            s = fmaf(a[k], w[k], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_xor(s, off, 64);
    if (lane == 0) {
        float r = -bias[emission[p]];
        r       = r - s;
        scores[p] = r;
    }
}

// ---- Nn::NeuralNetworkForwardNode's top layer (Nn/NeuralNetworkForwardNode.cc:140-160: the node emits network_.getTopLayerOutput()).
// The scorers keep LinearAndSoftmaxLayer's softmax switched off and negate; the node keeps it on (evaluate-softmax, default true,
// Nn/LinearAndActivationLayer.cc:85-106).  Math::FastMatrix<f32>::softmax per frame (Math/FastMatrix.hh:818-834): maximum of the
// column, x + (-1 * max), exp() = mt_vr_exp (::exp(double) narrowed, Math/FastVectorOperations.hh:57-63), the SEQUENTIAL f32 sum of
// FastVector::addSummedRows (Math/FastVector.hh:481-489), scal by (f32)1.0 / sum.
// top_negmax_kernel: one workgroup per frame; a = -score (the activation, exact), row maximum, e = exp(a - max) written in place.
__global__ __launch_bounds__(256) void top_negexp_kernel(float* __restrict__ x, int n, float* __restrict__ g_max) {
    __shared__ float s_red[4];
    float*           row = x + (size_t)blockIdx.x * n;
    float            mx  = -__builtin_inff();
    for (int i = threadIdx.x; i < n; i += 256)
        mx = fmaxf(mx, -row[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0)
        s_red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    const float value = -1.f * mx;  // addToAllRows(tmp, -1): alpha * max, then elem + value
    for (int i = threadIdx.x; i < n; i += 256) {
        const float a = -row[i];
        row[i]        = (float)exp((double)(a + value));
    }
    if (threadIdx.x == 0)
        g_max[blockIdx.x] = mx;
}

// lane = frame: the reference's sequential f32 sum over the outputs, in output order; 64 x 64 tiles go through LDS so that the global
// reads stay coalesced
__global__ __launch_bounds__(64) void top_rowsum_kernel(const float* __restrict__ x, int T, int n, float* __restrict__ g_sum) {
    __shared__ float s_tile[64][65];
    const int        lane = threadIdx.x, t0 = blockIdx.x * 64;
    float            acc = 0.f;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int w = min(64, n - c0);
        for (int r = 0; r < 64; ++r)
            s_tile[r][lane] = (t0 + r < T && lane < w) ? x[(size_t)(t0 + r) * n + c0 + lane] : 0.f;
        __syncthreads();
        for (int c = 0; c < w; ++c)
            acc = acc + s_tile[lane][c];
        __syncthreads();
    }
    if (t0 + lane < T)
        g_sum[t0 + lane] = acc;
}

// elementwise tail: LINEAR negates the scores (activation = -score, exact), SOFTMAX multiplies by (f32)1.0 / sum (Math::scal)
__global__ __launch_bounds__(256) void top_finish_kernel(float* __restrict__ x, long long total, int n, const float* __restrict__ g_sum) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        if (g_sum) {
            const float r = 1.f / g_sum[i / n];
            x[i]          = x[i] * r;
        }
        else
            x[i] = -x[i];
    }
}

// Nn::PrecomputedFeatureScorer::calculateScore (Nn/FeatureScorer.cc:291-310): the features ARE the network outputs:
// score(e) = -x[out(e)] + alpha * logPrior[out(e)], Core::Type<f32>::max for a disregarded class.  HBM bound: one read, one write.
__global__ __launch_bounds__(256) void precomputed_score_kernel(const float* __restrict__ x, int ldx, int T, int n_classes,
                                                               const int* __restrict__ class_to_output, const float* __restrict__ log_prior,
                                                               float prior_scale, float* __restrict__ scores) {
    const long long n = (long long)T * n_classes;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int t = (int)(i / n_classes), e = (int)(i - (long long)t * n_classes);
        const int o = class_to_output ? class_to_output[e] : e;
        float     sc = 3.402823466e+38f;
        if (o >= 0) {
            sc           = -x[(size_t)t * ldx + o];
            const float pr = prior_scale * log_prior[o];
            sc           = sc + pr;
        }
        scores[i] = sc;
    }
}

__global__ __launch_bounds__(256) void pack_input_f32(const float* __restrict__ x, int ldx, int T, int K, float* __restrict__ out,
                                                     int Kpad, int Tpad) {
    const long long n = (long long)Tpad * Kpad;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int t = (int)(i / Kpad), k = (int)(i - (long long)t * Kpad);
        out[i] = (t < T && k < K) ? x[(size_t)t * ldx + k] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 GEMM: D[n][t] = sum_k W[n][k] X[t][k];  W [Npad x Kpad], X [Tpad x ldx] bf16.
// hidden layers: out bf16 [Tpad x ldo] = act(D + bias);  last layer: out f32 [T x n_valid] = -(D + bias)
// Row strides and plane offsets (bf16 elements) of a GEMM launch.  Plain bf16: ldw = Kpad, every *lo = 0.  Split bf16: rows are
// [hi plane | lo plane]; wlo / xlo / olo are the columns where the lo planes of W, X and the hidden-layer output start.
struct GemmLd {
    int ldw, wlo, ldx, xlo, ldo, olo;
};

constexpr int BK      = 64;   // k per stage: one 128-byte LDS row per matrix row
constexpr int PAD_NT  = 256;  // N and T are padded to this (largest tile edge)

template<int BN_, int BT_, int WN_, int WT_, int STAGES_, int BK_ = 64, bool X3_ = false, int LW_ = 0>
struct GemmCfg {
    static constexpr int BN = BN_, BT = BT_, WN = WN_, WT = WT_, STAGES = STAGES_, BKC = BK_;
    // LW > 0 (round 6, gemm_bf16_kernel, hidden layers of small batches): LW extra LOADER waves issue every LDS-DMA piece and nothing
    // else; the NW computing waves never touch the address unit (six pieces of ~70 cycles per K-tile and wave stood in front of the
    // products of a 128 x 64 tile, one wave per SIMD and nothing to overlap them with -- gemm_mx_kernel's MxCfg::LW)
    static constexpr int LW = LW_;
    // X3: split-bf16 tiles.  Every operand tile is staged as TWO planes (hi = bf16(v), lo = bf16(v - hi)), the hi plane first,
    // and a k-slab issues the three products hi.hi, lo.hi, hi.lo from ONE set of fragment reads (4 planes per 3 products).
    static constexpr bool X3     = X3_;
    static constexpr int  PLANES = X3 ? 2 : 1;
    static constexpr int NW = WN * WT, THREADS = (NW + LW_) * 64;
    static constexpr int IW = LW_ > 0 ? LW_ : NW;           // waves that issue the LDS-DMA
    static constexpr int ROW_BYTES = BKC * 2;               // one LDS row per matrix row: 128 B (BK 64) or 64 B (BK 32)
    static constexpr int ROWS_PER_LOAD = 1024 / ROW_BYTES;  // rows covered by one wave-wide 16-byte global_load_lds
    static constexpr int CHUNKS = ROW_BYTES / 16;
    static constexpr int A_PLANE = BN * ROW_BYTES, B_PLANE = BT * ROW_BYTES;
    static constexpr int A_BYTES = PLANES * A_PLANE, B_BYTES = PLANES * B_PLANE, STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int A_LOADS = A_BYTES / 1024 / IW, B_LOADS = B_BYTES / 1024 / IW, LOADS = A_LOADS + B_LOADS;
    static constexpr int MI = BN / WN / 32, MJ = BT / WT / 32;  // 32x32 tiles per wave
    static_assert(BKC == 64 || BKC == 32, "BK must be 32 or 64");
    static_assert(!X3 || BKC == 32, "split-bf16 tiles are 32 k wide (four planes per stage)");
    static_assert(BN % (ROWS_PER_LOAD * IW) == 0 && BT % (ROWS_PER_LOAD * IW) == 0, "staging needs whole row groups per wave");
    static_assert(BN % (WN * 32) == 0 && BT % (WT * 32) == 0, "wave tile must be a multiple of 32x32");
    // XOR swizzle of the 16-byte chunks of a row so that ds_read_b128 of a 32-row MFMA fragment is conflict free:
    // 128-byte rows: chunk ^ ((row>>1)&7); 64-byte rows: chunk ^ ((row>>2)&3)
    __host__ __device__ static constexpr int xor_term(int r) { return BKC == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
    __host__ __device__ static constexpr int swz(int r, int c) { return r * ROW_BYTES + ((c ^ xor_term(r)) << 4); }
};

// byte offset of 16-byte chunk `c` (0..7) of row `r` inside a [rows][64] bf16 tile
__device__ __forceinline__ int swz(int r, int c) {
    return r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
}

template<int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Epilogue of the bf16 GEMM kernels: bias (+ activation, bf16 pack) or bias - prior, negate, and the per-tile arg-min
// partials of the output layer.  The MFMA accumulator layout gives a lane 4 consecutive outputs of ONE frame, i.e. a
// wave-wide store would touch 32 different output rows with 16/32-byte pieces.  Instead every wave transposes its
// tile through a private, padded LDS region (32 frames x 128 outputs at a time) and writes whole contiguous row
// segments (512 B per frame for f32 scores, 256 B for bf16 activations) with 16-byte lanes.
template<class C, bool LAST>
struct EpiCfg {
    static constexpr int WNR  = C::BN / C::WN;                // outputs (n) per wave
    static constexpr int WTT  = C::BT / C::WT;                // frames (t) per wave
    static constexpr int ELT  = LAST ? 4 : 2;                 // bytes per stored element
    static constexpr int ROWB = WNR * ELT + 16;               // padded LDS row: conflict-free b128 writes
    static constexpr int WAVE_BYTES = 32 * ROWB;
    static constexpr int STAGE_BYTES = C::NW * WAVE_BYTES;
    static constexpr int BEST_OFF = STAGE_BYTES;                                      // arg-min exchange [2][WN][BT]
    static constexpr int BYTES    = STAGE_BYTES + (LAST ? 2 * C::WN * C::BT * 4 : 0);
};

// LDS carve-up: [max(stage buffers, epilogue scratch)] [bias of the tile: BN floats]
template<class C, bool LAST>
__host__ __device__ constexpr int gemm_scratch_bytes() {
    return C::LDS_BYTES > EpiCfg<C, LAST>::BYTES ? C::LDS_BYTES : EpiCfg<C, LAST>::BYTES;
}

// Tile configurations whose accumulators live in the AGPR half of the register file (one wave per SIMD, 256 accumulator registers:
// MxCfg::PIPE == 2).  Left alone, the allocator copies all 256 of them into VGPRs behind the K loop -- and spills whatever else is alive.
// The epilogues pin one 32 x 32 block at a time ("a" constraint) right in front of its use: sixteen registers cross over at a time.
template<class C, class = void>
struct AccInAgprs : std::false_type {};
template<class C>
struct AccInAgprs<C, std::void_t<decltype(C::PIPE)>> : std::integral_constant<bool, C::PIPE == 2> {};
template<class C>
__device__ __forceinline__ void pin_block(f32x16& a) {
    if constexpr (AccInAgprs<C>::value)
        asm volatile("" : "+a"(a));
}

template<class C, int ACT, bool LAST, bool NOSTORE = false>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[C::MI][C::MJ], char* lds, const float* s_bias, void* __restrict__ out,
                                              int ldo, int olo, int n_valid, int t_valid, int n0, int t0, int tile_n, int wn, int wt, int lane, int tid,
                                              float* __restrict__ part_min, unsigned* __restrict__ part_idx, int part_ld) {
    using E = EpiCfg<C, LAST>;
    const int  wave      = tid >> 6;
    const bool want_best = LAST && part_min != nullptr;
    __syncthreads();  // every wave is done with the last K-tile: the stage buffers become epilogue scratch
    char*     w_lds = lds + wave * E::WAVE_BYTES;
    float*    s_min = (float*)(lds + E::BEST_OFF);                    // [WN][BT]
    unsigned* s_idx = (unsigned*)(lds + E::BEST_OFF + C::WN * C::BT * 4);
    const int tl32 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int j = 0; j < C::MJ; ++j) {
      // split-bf16 hidden layers leave as two planes per row, [hi | lo] (lo at column olo): one pass per plane
#pragma unroll
      for (int pl = 0; pl < (LAST ? 1 : C::PLANES); ++pl) {
        float    bmin = 3.402823466e+38f;
        unsigned bidx = 0xffffffffu;
        // ---- registers -> LDS [frame][output]
#pragma unroll
        for (int i = 0; i < C::MI; ++i) {
            if (pl == 0)
                pin_block<C>(acc[i][j]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = i * 32 + 8 * g + 4 * hh;  // output index inside the wave tile
                const int n  = n0 + wn * E::WNR + nl;
                float     v[4];
                const float4 b4 = *(const float4*)(s_bias + wn * E::WNR + nl);  // bias of this tile, staged in LDS
                const float  bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = acc[i][j][g * 4 + e] + bv[e];  // addToAllColumns
                if (LAST) {
                    *(float4*)(w_lds + tl32 * E::ROWB + nl * 4) = make_float4(-v[0], -v[1], -v[2], -v[3]);
                    if (want_best) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float sc = -v[e];
                            if (n + e < n_valid && sc < bmin) {  // ascending n within the lane
                                bmin = sc;
                                bidx = (unsigned)(n + e);
                            }
                        }
                    }
                }
                else {
                    const float a0 = activate<ACT>(v[0]), a1 = activate<ACT>(v[1]), a2 = activate<ACT>(v[2]), a3 = activate<ACT>(v[3]);
                    uint2       pk;
                    pk.x = pack_bf16(a0, a1);
                    pk.y = pack_bf16(a2, a3);
                    if (pl == 1) {  // lo = bf16(v - hi)
                        const unsigned h0 = pk.x, h1 = pk.y;
                        pk.x = pack_bf16(a0 - __uint_as_float(h0 << 16), a1 - __uint_as_float(h0 & 0xffff0000u));
                        pk.y = pack_bf16(a2 - __uint_as_float(h1 << 16), a3 - __uint_as_float(h1 & 0xffff0000u));
                    }
                    *(uint2*)(w_lds + tl32 * E::ROWB + nl * 2) = pk;
                }
            }
        }
        if (want_best) {
            // lanes l and l+32 hold the same frame, interleaved n: smaller index wins ties
            const float    om = __shfl_xor(bmin, 32, 64);
            const unsigned oi = (unsigned)__shfl_xor((int)bidx, 32, 64);
            if (om < bmin || (om == bmin && oi < bidx)) {
                bmin = om;
                bidx = oi;
            }
            if (lane < 32) {
                const int tl = wt * E::WTT + j * 32 + tl32;
                s_min[wn * C::BT + tl] = bmin;
                s_idx[wn * C::BT + tl] = bidx;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- LDS -> global: whole row segments, 16 bytes per lane
        const int tbase = t0 + wt * E::WTT + j * 32;
        const int nbase = n0 + wn * E::WNR;
        if (LAST) {
            constexpr int LPR = E::WNR / 4;   // lanes per row (float4 each)
            constexpr int RPI = 64 / LPR;     // rows per wave-wide store
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int    row = it * RPI + lane / LPR, c4 = lane % LPR;
                const float4 v   = *(const float4*)(w_lds + row * E::ROWB + c4 * 16);
                const int    t = tbase + row, n = nbase + 4 * c4;
                if (t < t_valid && !(NOSTORE && v.x != 123.456f)) {
                    float* o = (float*)out + (size_t)t * ldo + n;
                    if (n + 3 < n_valid && ((ldo & 3) == 0))
                        amx::nt_store(f32x4{v.x, v.y, v.z, v.w}, (f32x4*)o);  // 1.3 GB of scores per pass must not push the operand panels out of L2
                    else {
                        if (n < n_valid) o[0] = v.x;
                        if (n + 1 < n_valid) o[1] = v.y;
                        if (n + 2 < n_valid) o[2] = v.z;
                        if (n + 3 < n_valid) o[3] = v.w;
                    }
                }
            }
        }
        else {
            constexpr int LPR = E::WNR / 8;   // lanes per row (8 bf16 each)
            constexpr int RPI = 64 / LPR;
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int   row = it * RPI + lane / LPR, c8 = lane % LPR;
                const uint4 v   = *(const uint4*)(w_lds + row * E::ROWB + c8 * 16);
                if (!(NOSTORE && v.x != 0x12345678u))
                    *(uint4*)((bf16_t*)out + (size_t)(tbase + row) * ldo + pl * olo + nbase + 8 * c8) = v;  // padded buffer: no guards
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (want_best) {
        __syncthreads();
        for (int tl = tid; tl < C::BT; tl += C::THREADS) {
            float    bmin = s_min[tl];
            unsigned bidx = s_idx[tl];
#pragma unroll
            for (int w = 1; w < C::WN; ++w) {  // ascending n ranges: strict '<' keeps the first minimum
                const float m = s_min[w * C::BT + tl];
                if (m < bmin) {
                    bmin = m;
                    bidx = s_idx[w * C::BT + tl];
                }
            }
            part_min[(size_t)tile_n * part_ld + t0 + tl] = bmin;
            part_idx[(size_t)tile_n * part_ld + t0 + tl] = bidx;
        }
    }
}

}  // namespace amx
#include "ffnn_mx.hpp"
namespace amx {

// ablation / trace hooks (tools/gemm_probe.hip only; the library instantiates VAR = 0 and never touches them):
//   VAR & 8 no MFMA, 16 no operand loads after the first K-tile, 64 operand streaming only, 128 no epilogue, 256 epilogue
//   without global stores
//   VAR & 512  per-tile time stamps  g_gemm_trace[(block * 64 + step) * 4 + {start, k-loop end, epilogue end, XCC id}]
//   VAR & 1024 the workgroups of one XCD start every tile together (bounded spin on g_gemm_sync[xcd * 64 + step])
__device__ unsigned long long* g_gemm_trace = nullptr;
__device__ unsigned*           g_gemm_sync  = nullptr;

template<class C, int ACT, bool LAST, int VAR>
__global__ __launch_bounds__(C::THREADS) void gemm_bf16_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ X,
                                                              const float* __restrict__ bias, void* __restrict__ out, int Kpad, GemmLd ld,
                                                              int n_valid, int t_valid, int n_tiles_n, int n_tiles_total, int GT, int GN,
                                                              float* __restrict__ part_min,
                                                              unsigned* __restrict__ part_idx, int part_ld) {
    const int ldx = ld.ldx, ldo = ld.ldo;
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [STAGES][W tile | X tile]
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn   = wave / C::WT, wt = wave % C::WT;

    // XCD-aware tile order.  Workgroup b runs on XCD b%8 (observed placement, speed only); the CUs of
    // an XCD pick up their workgroups in increasing b/8.  Give every XCD a contiguous range of a virtual
    // order v in which 32 consecutive tiles (= the 32 CUs of the XCD at one time) form a 4(t) x 8(n)
    // super-tile: per K-step the XCD's L2 then fetches 4 X slabs + 8 W slabs for 32 tiles instead of
    // 1 + 32, and neighbouring super-tiles keep sharing the X panel.
    // Persistent workgroups: the grid covers the CUs once and every workgroup walks a strided list of tiles, so
    // the epilogue stores of one tile drain while the next tile's operands stream in (no workgroup turnaround).
  for (int vi = blockIdx.x; vi < n_tiles_total; vi += gridDim.x) {
    const int nwg = n_tiles_total;
    int       tile_t, tile_n;
    {
        const int b = vi;  // vi % 8 == blockIdx.x % 8 (grid is a multiple of 8): same XCD for all tiles of a workgroup
        const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, k = b >> 3;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;  // bijective for any nwg
        const int n_tiles_t = nwg / n_tiles_n;
        if (GT > 0 && GN > 0) {
            // bands of GT frame-tiles; inside a band the tiles are walked in column blocks of GN: the 32 tiles an XCD
            // works on at one time form (up to edge effects) a GT x GN block = GT X panels + GN W panels per K-step
            const int band  = v / (GT * n_tiles_n), w = v - band * (GT * n_tiles_n);
            const int bt0   = band * GT, bh = min(GT, n_tiles_t - bt0);  // last band may be shorter
            const int blk   = w / (bh * GN), u = w - blk * (bh * GN);
            const int bn0   = blk * GN, bw = min(GN, n_tiles_n - bn0);   // last column block may be narrower
            tile_t          = bt0 + u / bw;
            tile_n          = bn0 + u % bw;
        }
        else {
            tile_t = v / n_tiles_n;
            tile_n = v - tile_t * n_tiles_n;
        }
    }
    const int n0 = tile_n * C::BN, t0 = tile_t * C::BT;
    const int step = (vi - (int)blockIdx.x) / (int)gridDim.x;
    if (VAR & 1024) {
        if (tid == 0) {
            const int xcd  = blockIdx.x & 7;
            const int lo   = step * (int)gridDim.x, hi = min(lo + (int)gridDim.x, n_tiles_total);
            const int want = (hi - lo + 7 - xcd) / 8;  // tiles of this step with index % 8 == xcd (lo % 8 == 0)
            unsigned* c    = g_gemm_sync + xcd * 64 + (step & 63);
            __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int spin = 0; spin < 20000; ++spin) {
                if ((int)__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want)
                    break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
    if ((VAR & 512) && tid == 0) {
        g_gemm_trace[((size_t)blockIdx.x * 64 + (step & 63)) * 4 + 0] = wall_clock64();
        g_gemm_trace[((size_t)blockIdx.x * 64 + (step & 63)) * 4 + 3] = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID
    }

    // ---- staging: one global_load_lds moves 8 rows x 128 B per wave.  LDS linear position of this
    // lane's 16 B for instruction i: pos = (i*NW + wave)*1024 + lane*16 -> row = pos/128, physical
    // chunk = (pos%128)/16; it must hold logical chunk = phys ^ ((row>>1)&7) of that row.
    constexpr bool HAS_LW = C::LW > 0;
    const bool     loader = HAS_LW && wave >= C::NW;                 // issues the LDS-DMA, takes part in the barriers, computes nothing
    const bool     issuer = !HAS_LW || loader;
    const int      iwave  = HAS_LW ? (loader ? wave - C::NW : 0) : wave;  // index among the issuing waves
    const bf16_t* gW[C::A_LOADS];
    const bf16_t* gX[C::B_LOADS];
#pragma unroll
    for (int i = 0; i < C::A_LOADS; ++i) {
        const int pos = (i * C::IW + iwave) * 1024 + lane * 16;
        const int plane = pos / C::A_PLANE, rpos = pos % C::A_PLANE;  // split bf16: hi plane, then lo plane
        const int row = rpos / C::ROW_BYTES, phys = (rpos % C::ROW_BYTES) >> 4;
        gW[i]         = W + (size_t)(n0 + row) * ld.ldw + plane * ld.wlo + (phys ^ C::xor_term(row)) * 8;
    }
#pragma unroll
    for (int i = 0; i < C::B_LOADS; ++i) {
        const int pos = (i * C::IW + iwave) * 1024 + lane * 16;
        const int plane = pos / C::B_PLANE, rpos = pos % C::B_PLANE;
        const int row = rpos / C::ROW_BYTES, phys = (rpos % C::ROW_BYTES) >> 4;
        gX[i]         = X + (size_t)(t0 + row) * ldx + plane * ld.xlo + (phys ^ C::xor_term(row)) * 8;
    }
    auto stage = [&](int slot, int kt) {
        if (!issuer)
            return;
        char* base = lds + slot * C::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < C::A_LOADS; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(gW[i] + (size_t)kt * C::BKC),
                                             (__attribute__((address_space(3))) void*)(base + (i * C::IW + iwave) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < C::B_LOADS; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(gX[i] + (size_t)kt * C::BKC),
                                             (__attribute__((address_space(3))) void*)(base + C::A_BYTES + (i * C::IW + iwave) * 1024), 16, 0, 0);
    };

    f32x16 acc[C::MI][C::MJ];
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int j = 0; j < C::MJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;
    // the tile's bias vector goes to LDS now (visible after the K-loop's barriers): per-lane global bias loads in
    // the epilogue were a chain of ~30 dependent L2 round trips per tile
    float* s_bias = (float*)(lds + gemm_scratch_bytes<C, LAST>());
    for (int i = tid; i < C::BN; i += C::THREADS)
        s_bias[i] = bias[n0 + i];

    // ---- software pipeline, ONE barrier per K-tile:
    //   wait(my loads of tile kt) ; barrier (=> everybody's loads of kt landed AND everybody finished
    //   reading tile kt-1) ; issue tile kt+STAGES-1 into the slot tile kt-1 occupied ; multiply tile kt.
    // STAGES-1 tiles stay in flight across the barrier (counted vmcnt, never a drain).
    const int KT = Kpad / C::BKC;
    const int frow = lane & 31;
    const int fk   = lane >> 5;  // which 8-element half of a 16-wide k slab
    {
#pragma unroll
    for (int s = 0; s < C::STAGES - 1; ++s)
        if (s < KT)
            stage(s, s);
    for (int kt = 0; kt < KT; ++kt) {
        // tiles kt .. min(kt+STAGES-2, KT-1) are outstanding; keep all but tile kt in flight
        const int ahead = min(C::STAGES - 2, KT - 1 - kt);
        static_assert(C::STAGES <= 6 && 4 * C::LOADS < 64, "vmcnt immediate");
        if (!issuer)
            ;  // a computing wave beside loader waves has no vector-memory operation in flight inside the loop
        else if (C::STAGES >= 6 && ahead == 4)
            wait_vmcnt<4 * C::LOADS>();
        else if (C::STAGES >= 5 && ahead == 3)
            wait_vmcnt<3 * C::LOADS>();
        else if (C::STAGES >= 4 && ahead == 2)
            wait_vmcnt<2 * C::LOADS>();
        else if (C::STAGES >= 3 && ahead == 1)
            wait_vmcnt<C::LOADS>();
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const bool more  = kt + C::STAGES - 1 < KT;
        const int  nslot = (kt + C::STAGES - 1) % C::STAGES;
        const char* wbase = lds + (kt % C::STAGES) * C::STAGE_BYTES;
        const char* xbase = wbase + C::A_BYTES;
        // Small tiles (128 x 64, plain bf16; round 6): ALL fragment reads of the K-tile go out as one burst in front of the refill's
        // LDS-DMA pieces and of the first product -- the loop below pays the LDS round trip once per 16-k slab (four times per
        // K-tile) and the pieces' issue time (six per wave, ~70 cycles each) in front of the first read, one wave per SIMD and nothing
        // to overlap either with (the lesson of gemm_mx_kernel's one-tile-per-CU loop).  Same products, same
        // order per accumulator: bit-identical.
        constexpr bool BURST = C::BN == 128 && C::BT == 64 && !C::X3 && (VAR & (8 | 64)) == 0;
        if constexpr (BURST) {
            constexpr int KS = C::BKC / 16;
            if (loader) {   // (only configurations with loader waves take this path with them: static_assert below)
                if (more && !((VAR & 16) && kt > 0))
                    stage(nslot, kt + C::STAGES - 1);
                continue;
            }
            bf16x8 a[KS][C::MI], b[KS][C::MJ];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < C::MI; ++i)
                    a[ks][i] = *(const bf16x8*)(wbase + C::swz(wn * (C::BN / C::WN) + i * 32 + frow, ks * 2 + fk));
#pragma unroll
                for (int j = 0; j < C::MJ; ++j)
                    b[ks][j] = *(const bf16x8*)(xbase + C::swz(wt * (C::BT / C::WT) + j * 32 + frow, ks * 2 + fk));
            }
            if (more && !((VAR & 16) && kt > 0))
                stage(nslot, kt + C::STAGES - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < C::MI; ++i)
#pragma unroll
                    for (int j = 0; j < C::MJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
        }
        static_assert(!HAS_LW || BURST, "loader waves: the small plain-bf16 tile only");
        if (more && !((VAR & 16) && kt > 0))  // VAR & 16: ablation, no operand loads after the first K-tile
            stage(nslot, kt + C::STAGES - 1);
        if (VAR & 64) {
            // ablation: operand streaming only (no fragment reads, no MFMA)
        }
        else {
#pragma unroll
            for (int ks = 0; ks < C::BKC / 16; ++ks) {
                bf16x8 a[C::MI], b[C::MJ];
#pragma unroll
                for (int i = 0; i < C::MI; ++i)
                    a[i] = *(const bf16x8*)(wbase + C::swz(wn * (C::BN / C::WN) + i * 32 + frow, ks * 2 + fk));
#pragma unroll
                for (int j = 0; j < C::MJ; ++j)
                    b[j] = *(const bf16x8*)(xbase + C::swz(wt * (C::BT / C::WT) + j * 32 + frow, ks * 2 + fk));
                if (C::X3) {
                    // one set of fragment reads, three products: hi.hi, lo.hi, hi.lo -- in this order in EVERY split-bf16 kernel, so
                    // that all tile configurations accumulate identically
                    bf16x8 al[C::MI], bl[C::MJ];
#pragma unroll
                    for (int i = 0; i < C::MI; ++i)
                        al[i] = *(const bf16x8*)(wbase + C::A_PLANE + C::swz(wn * (C::BN / C::WN) + i * 32 + frow, ks * 2 + fk));
#pragma unroll
                    for (int j = 0; j < C::MJ; ++j)
                        bl[j] = *(const bf16x8*)(xbase + C::B_PLANE + C::swz(wt * (C::BT / C::WT) + j * 32 + frow, ks * 2 + fk));
#pragma unroll
                    for (int i = 0; i < C::MI; ++i)
#pragma unroll
                        for (int j = 0; j < C::MJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < C::MI; ++i)
#pragma unroll
                        for (int j = 0; j < C::MJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < C::MI; ++i)
#pragma unroll
                        for (int j = 0; j < C::MJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bl[j], acc[i][j], 0, 0, 0);
                    continue;
                }
#pragma unroll
                for (int i = 0; i < C::MI; ++i)
#pragma unroll
                    for (int j = 0; j < C::MJ; ++j) {
                        if (VAR & 8) {  // ablation: keep the fragment reads alive, skip the matrix pipe
                            asm volatile("" ::"v"(a[i]), "v"(b[j]));
                        }
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        // this wave's LDS reads have returned before it can arrive at the next barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    }

    if ((VAR & 512) && tid == 0)
        g_gemm_trace[((size_t)blockIdx.x * 64 + (step & 63)) * 4 + 1] = wall_clock64();
    if (VAR & 128) {  // ablation: no epilogue (a reduction over ALL accumulators keeps every MFMA alive)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::MJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sum += acc[i][j][r];
        if (sum == 123.456f)
            ((float*)out)[tid] = sum;
    }
    else if (VAR & 256)
        gemm_epilogue<C, ACT, LAST, true>(acc, lds, s_bias, out, ldo, ld.olo, n_valid, t_valid, n0, t0, tile_n, wn, wt, lane, tid, part_min, part_idx, part_ld);
    else if (HAS_LW && loader) {
        static_assert(!(HAS_LW && LAST), "loader waves: hidden layers only (the score epilogue's second barrier and its arg-min exchange count NW waves)");
        __syncthreads();  // the epilogue's first barrier: every wave is done with the last K-tile
    }
    else
        gemm_epilogue<C, ACT, LAST>(acc, lds, s_bias, out, ldo, ld.olo, n_valid, t_valid, n0, t0, tile_n, wn, wt, lane, tid, part_min, part_idx, part_ld);
    __syncthreads();  // LDS (stages / arg-min scratch) is reused by the next tile
    if ((VAR & 512) && tid == 0)
        g_gemm_trace[((size_t)blockIdx.x * 64 + (step & 63)) * 4 + 2] = wall_clock64();
  }
}

// ---------------------------------------------------------------------------------------------
// gemm_bf16_pipe_kernel: the large-batch kernel (256x256x64 tiles, 8 waves of 128x64).  Same K-loop as above, but the
// operand stream runs ACROSS tiles and the epilogue is rebuilt around it:
//   * in the last K-step of a tile the free stage already receives the first K-tile of the workgroup's NEXT tile (and
//     its bias vector, by LDS-DMA), so those loads precede the epilogue's global stores in the wave's VMEM order.  The
//     next tile's first wait is then `s_waitcnt vmcnt(#stores)`: it waits for the loads, not for the score stores
//     (completion is reported in issue order on gfx9), which keep draining under the first K-steps.
//   * the epilogue transposes through the OTHER stage: 8 KB per wave (32 frames x 256 B, 16-byte chunks XOR-swizzled by
//     the row, conflict free for ds_write_b128/b64 and ds_read_b128 -- tools/lds_conflicts.py), f32 scores in four
//     sub-passes of 32 frames x 64 states, bf16 activations in two passes of 32 frames x 128 units.  Bias values are
//     fetched four float4 at a time, fragment reads and stores are batched, so a pass has ~6 waits instead of ~50.
//   * interior tiles store without guards (their store count is what the counted wait relies on); edge tiles use the
//     guarded path and fall back to a full drain.
template<class C, bool LAST>
struct PipeLds {
    static constexpr int WAVE_SCRATCH = 32 * 256;
    static constexpr int BEST_OFF     = 2 * C::STAGE_BYTES;                 // arg-min exchange [2][WN][BT]
    static constexpr int BEST_BYTES   = LAST ? 2 * C::WN * C::BT * 4 : 0;
    static constexpr int BIAS_OFF     = BEST_OFF + BEST_BYTES;              // [2][BN] f32, double buffered per tile
    static constexpr int BYTES        = BIAS_OFF + 2 * C::BN * 4;
    static constexpr int STORES       = LAST ? C::MJ * 2 * 8 : C::MJ * 8 * C::PLANES;   // global stores per wave and interior tile
    static_assert(C::STAGES == 2 && (C::X3 ? C::BKC == 32 : C::BKC == 64), "two stages of 64 KB: 2 x 256 rows x 64 k, or 4 planes x 256 rows x 32 k");
    static_assert(C::BN / C::WN == 128 && C::MI == 4, "wave tile is 128 outputs wide");
    static_assert(C::NW * WAVE_SCRATCH <= C::STAGE_BYTES, "epilogue scratch must fit into one stage");
    static_assert(C::BN * 4 == 1024, "the bias vector is one wave-wide 16-byte LDS-DMA");
    static_assert(STORES + 2 < 64, "vmcnt immediate");
};

__device__ __forceinline__ int pipe_swz(int row, int chunk) {
    return row * 256 + ((chunk ^ (row & 15)) << 4);
}

// fragment registers of half a K-tile of the pipelined kernel.  Plain bf16: two k-slabs of (W, X) fragments, 2 MI MJ MFMAs.  Split
// bf16: one k-slab of (W_hi, W_lo, X_hi, X_lo) fragments, 3 MI MJ MFMAs.  12 fragments = 48 registers either way.
template<class C>
struct PipeFrag {
    static constexpr int NS = C::X3 ? 1 : 2;
    bf16x8 a[NS][C::MI], b[NS][C::MJ], al[C::X3 ? C::MI : 1], bl[C::X3 ? C::MJ : 1];
};

template<class C, int H>
__device__ __forceinline__ void pipe_read_half(PipeFrag<C>& f, const char* stage, int a_row, int b_row, int fk) {
    const char* wbase = stage;
    const char* xbase = stage + C::A_BYTES;
#pragma unroll
    for (int s = 0; s < PipeFrag<C>::NS; ++s) {
        const int ks = H * PipeFrag<C>::NS + s;  // k-slab of the K-tile
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
            f.a[s][i] = *(const bf16x8*)(wbase + C::swz(a_row + i * 32, ks * 2 + fk));
#pragma unroll
        for (int j = 0; j < C::MJ; ++j)
            f.b[s][j] = *(const bf16x8*)(xbase + C::swz(b_row + j * 32, ks * 2 + fk));
        if (C::X3) {
#pragma unroll
            for (int i = 0; i < C::MI; ++i)
                f.al[i] = *(const bf16x8*)(wbase + C::A_PLANE + C::swz(a_row + i * 32, ks * 2 + fk));
#pragma unroll
            for (int j = 0; j < C::MJ; ++j)
                f.bl[j] = *(const bf16x8*)(xbase + C::B_PLANE + C::swz(b_row + j * 32, ks * 2 + fk));
        }
    }
}

// MFMAs M0 .. M1-1 of one half, in the order every bf16 GEMM kernel of this file uses (slab by slab; split bf16: hi.hi, lo.hi, hi.lo
// of the slab)
template<class C, int M0, int M1, int DBG>
__device__ __forceinline__ void pipe_mfma_range(f32x16 (&acc)[C::MI][C::MJ], const PipeFrag<C>& f) {
    constexpr int T = C::MI * C::MJ;
#pragma unroll
    for (int m = M0; m < M1; ++m) {
        const int grp = m / T, i = (m % T) / C::MJ, j = m % C::MJ;
        const bf16x8 av = C::X3 ? (grp == 1 ? f.al[i] : f.a[0][i]) : f.a[grp][i];
        const bf16x8 bv = C::X3 ? (grp == 2 ? f.bl[j] : f.b[0][j]) : f.b[grp][j];
        if (DBG & 8)  // ablation: fragment reads stay alive, the matrix pipe is idle
            asm volatile("" ::"v"(av), "v"(bv));
        else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i][j], 0, 0, 0);
    }
}

// DBG: tools/gemm_probe.hip only (1 time stamps, 2 no global stores, 4 no epilogue, 8 no MFMA, 16 no operand DMA after the prologue,
// 64 L2-resident operands)
template<class C, int ACT, bool LAST, int DBG = 0>
__global__ __launch_bounds__(C::THREADS) void gemm_bf16_pipe_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ X,
                                                                   const float* __restrict__ bias, void* __restrict__ out, int Kpad, GemmLd ld,
                                                                   int n_valid, int t_valid, int n_tiles_n, int n_tiles_total, int GT,
                                                                   int GN, int out_aligned, float* __restrict__ part_min,
                                                                   unsigned* __restrict__ part_idx, int part_ld) {
    using P = PipeLds<C, LAST>;
    const int ldx = ld.ldx, ldo = ld.ldo;
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [stage 0 | stage 1 | arg-min exchange | bias x 2]
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn   = wave / C::WT, wt = wave % C::WT;
    const int KT   = Kpad / C::BKC;
    const bool want_best = LAST && part_min != nullptr;

    auto coords = [&](int vi, int& tile_t, int& tile_n) {  // same XCD-aware order as gemm_bf16_kernel
        if (DBG & 64) {  // ablation: every workgroup streams the operands of ONE of 8 tiles (everything an L2 hit)
            tile_t = blockIdx.x & 7;
            tile_n = 0;
            return;
        }
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
    };
    // staging: piece i of a wave (one wave-wide 16-byte LDS-DMA = 1 KB) lands at LDS offset (i*NW + wave) KB of the operand's region.
    // 128-byte rows (bf16): a piece is 8 rows, a lane's 16 bytes sit at row lane/8, physical chunk lane%8, which must hold logical
    // chunk phys ^ ((row>>1)&7) = phys ^ xr (independent of i).  64-byte rows (split bf16): 16 rows per piece, row lane/4, physical
    // chunk lane%4 ^ ((row>>2)&3) = lane%4 ^ (lane>>4); the region is the hi plane followed by the lo plane, i.e. the wave's pieces
    // 0 .. A_LOADS/2-1 read the hi columns and the rest the same rows of the lo columns (wlo / xlo further right).
    // Addresses are a wave-uniform 64-bit base (SGPR pair: tile, K-tile, plane, piece and the wave's rows) plus ONE 32-bit lane offset per
    // operand (row lane/4 or lane/8 of the piece, swizzled chunk): the global_load_lds saddr form, two address VGPRs for the whole kernel.
    constexpr int  RPL  = C::ROWS_PER_LOAD;
    const int      lrow = C::X3 ? (lane >> 2) : (lane >> 3);
    const int      lchk = C::X3 ? ((lane & 3) ^ ((lane >> 4) & 3)) : ((lane & 7) ^ (((wave & 1) * 4 + (lane >> 4)) & 7));
    const unsigned voffW = (unsigned)(lrow * ld.ldw + (lchk << 3)) * 2u, voffX = (unsigned)(lrow * ldx + (lchk << 3)) * 2u;  // bytes
    const size_t   waveW = (size_t)wave * RPL * ld.ldw, waveX = (size_t)wave * RPL * ldx;
    const size_t   stepW = (size_t)C::NW * RPL * ld.ldw, stepX = (size_t)C::NW * RPL * ldx;
    constexpr int  PPA = C::A_LOADS / C::PLANES, PPB = C::B_LOADS / C::PLANES;  // pieces per plane and wave
    auto issue = [&](int slot, const bf16_t* pw, const bf16_t* px) {  // pw / px: wave-uniform, first row of the tile, first k of the K-tile
        char* base = lds + slot * C::STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < C::A_LOADS; ++i) {
            const bf16_t* u = pw + waveW + (size_t)(i / PPA) * ld.wlo + (size_t)(i % PPA) * stepW;
            __builtin_amdgcn_global_load_lds((const void*)((const char*)u + voffW), (__attribute__((address_space(3))) void*)(base + i * C::NW * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < C::B_LOADS; ++i) {
            const bf16_t* u = px + waveX + (size_t)(i / PPB) * ld.xlo + (size_t)(i % PPB) * stepX;
            __builtin_amdgcn_global_load_lds((const void*)((const char*)u + voffX),
                                             (__attribute__((address_space(3))) void*)(base + C::A_BYTES + i * C::NW * 1024), 16, 0, 0);
        }
    };
    auto issue_piece = [&](int i, int slot, const bf16_t* pw, const bf16_t* px) {  // piece i of the K-tile: 0 .. A_LOADS-1 W, then X
        char* base = lds + slot * C::STAGE_BYTES + wave * 1024;
        if (i < C::A_LOADS) {
            const bf16_t* u = pw + waveW + (size_t)(i / PPA) * ld.wlo + (size_t)(i % PPA) * stepW;
            __builtin_amdgcn_global_load_lds((const void*)((const char*)u + voffW), (__attribute__((address_space(3))) void*)(base + i * C::NW * 1024), 16, 0, 0);
        }
        else {
            const int q = i - C::A_LOADS;
            const bf16_t* u = px + waveX + (size_t)(q / PPB) * ld.xlo + (size_t)(q % PPB) * stepX;
            __builtin_amdgcn_global_load_lds((const void*)((const char*)u + voffX),
                                             (__attribute__((address_space(3))) void*)(base + C::A_BYTES + q * C::NW * 1024), 16, 0, 0);
        }
    };
    auto issue_bias = [&](int buf, int tile_n) {  // wave 0: BN floats = 64 lanes x 16 B
        if (wave == 0) {  // scalar base + 32-bit lane offset (saddr form: no 64-bit lane pointer kept alive across the K-loop)
            const unsigned long long ub = (unsigned long long)(bias + tile_n * C::BN);
            const char* sbase = (const char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ub >> 32)) << 32) |
                                              (unsigned)__builtin_amdgcn_readfirstlane((int)ub));
            __builtin_amdgcn_global_load_lds((const void*)(sbase + (unsigned)(lane * 16)),
                                             (__attribute__((address_space(3))) void*)(lds + P::BIAS_OFF + buf * C::BN * 4), 16, 0, 0);
        }
    };

    int vi = blockIdx.x;
    if (vi >= n_tiles_total)
        return;
    // ---- fetch cursor: the operand stream is ONE sequence of K-tiles across the workgroup's tiles; K-tile number c of the stream goes
    // to stage c & 1.  The cursor runs up to two K-tiles ahead of the arithmetic.
    int           c_vi = vi, c_kt = 0, c_buf = 0, c_tn = 0;
    unsigned      c       = 0;     // K-tiles issued so far
    bool          c_valid = true;  // the cursor points at an existing K-tile
    const bf16_t *c_pw, *c_px;
    {
        int tt;
        coords(c_vi, tt, c_tn);
        c_pw = W + (size_t)c_tn * C::BN * ld.ldw;
        c_px = X + (size_t)tt * C::BT * ldx;
    }
    // cursor_take: the K-tile at the cursor becomes (q_slot, q_pw, q_px), whose pieces are issued by issue_piece / issue_rest; the
    // cursor moves on
    const bf16_t *q_pw = nullptr, *q_px = nullptr;
    int           q_slot = 0;
    bool          q_on   = false;
    auto cursor_take = [&]() {
        q_on   = !((DBG & 16) && c >= 2);  // DBG 16: ablation, no operand traffic after the prologue
        q_slot = c & 1;
        q_pw   = c_pw;
        q_px   = c_px;
        if (q_on && c_kt == 0)
            issue_bias(c_buf, c_tn);  // the tile's bias vector travels in front of its first K-tile
        ++c;
        if (++c_kt < KT) {
            c_pw += C::BKC;
            c_px += C::BKC;
        }
        else {  // the stream continues with the workgroup's next tile
            c_kt = 0;
            c_buf ^= 1;
            c_vi += (int)gridDim.x;
            c_valid = c_vi < n_tiles_total;
            if (c_valid) {
                int tt;
                coords(c_vi, tt, c_tn);
                c_pw = W + (size_t)c_tn * C::BN * ld.ldw;
                c_px = X + (size_t)tt * C::BT * ldx;
            }
        }
    };
    auto issue_next = [&]() {  // a whole K-tile in one burst (prologue, first K-tile of a tile)
        cursor_take();
        if (q_on)
            issue(q_slot, q_pw, q_px);
    };
    issue_next();
    if (c_valid)
        issue_next();
    int      bias_buf = 0;
    unsigned g        = 0;      // running K-tile count of the arithmetic; K-tile g lives in stage g & 1
    bool     counted  = false;  // the loads in flight are followed by exactly P::STORES (+2) stores of this wave
    const int frow = lane & 31, fk = lane >> 5;
    const bool extra = LAST && want_best && wave * 64 < C::BT;  // waves that also write the arg-min partials
    const int  a_row = wn * (C::BN / C::WN) + frow, b_row = wt * (C::BT / C::WT) + frow;
    PipeFrag<C> S0, S1;  // fragment registers of the two halves of a K-tile

    // K-loop schedule (one K-tile = two halves of MPH MFMAs each; S0 / S1 hold the halves' fragments):
    //     read S1 <- stage(g) | MFMA S0 (first part) | lgkmcnt(0), barrier R(g): everybody has stage(g) in registers | DMA(g+2) -> stage(g)
    //     | MFMA S0 (rest), MFMA S1 (first part) | vmcnt(younger loads), barrier X(g+1): K-tile g+1 has landed | read S0 <- stage(g+1)
    //     | MFMA S1 (rest)
    // i.e. a stage is occupied only from its landing to the moment its two halves sit in registers, the LDS-DMA of K-tile g+2 has one and
    // a half K-tiles of MFMA time to land (the two-stage loop with one barrier per K-tile gave it at most one; it measured 1.9 us per
    // 64 KB K-tile against 1.7 us of MFMA work), and every fragment read is issued half a K-tile before its first use.
    // At the last K-tile of a tile nothing is issued at R(g): the freed stage is the epilogue's transposition scratch; that K-tile of the
    // stream is issued behind the next tile's first barrier instead.
    for (;;) {
        f32x16 acc[C::MI][C::MJ];
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::MJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][r] = 0.f;
        int tile_t, tile_n;
        coords(vi, tile_t, tile_n);
        const int  n0 = tile_n * C::BN, t0 = tile_t * C::BT, cur_tile_n = tile_n;
        const int  nvi      = vi + (int)gridDim.x;
        const bool has_next = nvi < n_tiles_total;
        const int  step     = (vi - (int)blockIdx.x) / (int)gridDim.x;
        if ((DBG & 1) && tid == 0) {
            g_gemm_trace[((size_t)blockIdx.x * 64 + (step & 63)) * 4 + 0] = wall_clock64();
            g_gemm_trace[((size_t)blockIdx.x * 64 + (step & 63)) * 4 + 3] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
        }
        // ---- X(g) of the tile's first K-tile: its loads are followed by the previous tile's stores (counted) or, in the very first
        // tile, by the second K-tile of the stream
        if (g == 0) {
            if (c >= 2)
                wait_vmcnt<C::LOADS>();
            else
                wait_vmcnt<0>();
        }
        else if (counted) {
            if (extra)
                wait_vmcnt<P::STORES + 2>();
            else
                wait_vmcnt<P::STORES>();
        }
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // also: every wave is through with the previous tile's epilogue scratch
        if (c == g + 1 && c_valid)
            issue_next();  // the K-tile that was held back at the previous tile's last R
        pipe_read_half<C, 0>(S0, lds + (g & 1) * C::STAGE_BYTES, a_row, b_row, fk);

        for (int kt = 0; kt < KT; ++kt, ++g) {
            constexpr int MPH = (C::X3 ? 3 : 2) * C::MI * C::MJ, QN = MPH / 4;
            static_assert(MPH % 4 == 0 && C::LOADS == 8, "eight pieces per wave and K-tile, between quarters of a half");
            const bool  last = kt + 1 == KT;
            const char* cur  = lds + (g & 1) * C::STAGE_BYTES;
            const char* nxt  = lds + ((g + 1) & 1) * C::STAGE_BYTES;
            // The 8 LDS-DMA pieces of K-tile g+2 go out one at a time between groups of MFMAs: a burst of 64 pieces per workgroup behind
            // the barrier parks every wave in front of the address unit, and a wave issues in order -- no MFMA leaves while its load
            // waits (burst -> spread: output layer 3.53 -> 3.19 ms in split bf16, 1.37 -> 1.27 ms in bf16).  Measured and dropped: the
            // two waves of a SIMD handing their pieces over half an interval apart (3.25 ms).
#define AMX_PIECE(I)                                       \
    do {                                                   \
        __builtin_amdgcn_sched_barrier(0);                 \
        if (on)                                            \
            issue_piece(I, q_slot, q_pw, q_px);            \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)
            pipe_read_half<C, 1>(S1, cur, a_row, b_row, fk);
            pipe_mfma_range<C, 0, 2 * QN, DBG>(acc, S0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // R(g)
            const bool iss = !last && c_valid;
            if (iss)
                cursor_take();
            const bool on = iss && q_on;
            AMX_PIECE(0);
            pipe_mfma_range<C, 2 * QN, 3 * QN, DBG>(acc, S0);
            AMX_PIECE(1);
            pipe_mfma_range<C, 3 * QN, 4 * QN, DBG>(acc, S0);
            AMX_PIECE(2);
            pipe_mfma_range<C, 0, QN, DBG>(acc, S1);
            AMX_PIECE(3);
            pipe_mfma_range<C, QN, 2 * QN, DBG>(acc, S1);
            AMX_PIECE(4);
            if (!last) {
                if (on)
                    wait_vmcnt<5>();  // K-tile g+1 has landed; the five pieces of K-tile g+2 stay in flight
                else
                    wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();  // X(g+1)
                pipe_read_half<C, 0>(S0, nxt, a_row, b_row, fk);
            }
            AMX_PIECE(5);
            pipe_mfma_range<C, 2 * QN, 3 * QN, DBG>(acc, S1);
            AMX_PIECE(6);
            pipe_mfma_range<C, 3 * QN, 4 * QN, DBG>(acc, S1);
            AMX_PIECE(7);
#undef AMX_PIECE
        }

        if ((DBG & 1) && tid == 0)
            g_gemm_trace[((size_t)blockIdx.x * 64 + (step & 63)) * 4 + 1] = wall_clock64();
        if (DBG & 4) {  // ablation: no epilogue (a reduction over all accumulators keeps the MFMAs alive)
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < C::MI; ++i)
#pragma unroll
                for (int j = 0; j < C::MJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sum += acc[i][j][r];
            if (sum == 123.456f)
                ((float*)out)[tid] = sum;
            counted = false;
            if (!has_next)
                break;
            vi = nvi;
            bias_buf ^= 1;
            continue;
        }
        // ------------------------------------------------------------------ epilogue of tile (t0, n0)
        // every wave is through R of the last K-tile, i.e. done reading stage (g-1)&1, and nothing has been issued into it: that
        // stage is the transposition scratch
        // the epilogue's lane-dependent addresses are derived from an opaque copy of the lane id: computed from `lane` they are
        // loop invariants of the tile loop, get hoisted in front of the K-loop, spilled there (the K-loop owns the register file) and
        // reloaded between the stores behind s_waitcnt vmcnt(0)
        int elane = lane;
        asm volatile("" : "+v"(elane));
        const int    tl32 = elane & 31, hh = elane >> 5;
        char*        w_lds = lds + ((g - 1) & 1) * C::STAGE_BYTES + wave * P::WAVE_SCRATCH;
        const float* sb    = (const float*)(lds + P::BIAS_OFF + bias_buf * C::BN * 4) + wn * 128;
        const int    nbase = n0 + wn * 128;
        const bool   interior = LAST ? (t0 + C::BT <= t_valid && n0 + C::BN <= n_valid && out_aligned) : true;
        const int    prow = elane >> 4, pc = elane & 15;  // phase 2: 4 rows x 16 chunks per wave-wide access
        if (LAST) {
            float*    s_min = (float*)(lds + P::BEST_OFF);
            unsigned* s_idx = (unsigned*)(lds + P::BEST_OFF + C::WN * C::BT * 4);
            const bool edge_n = n0 + C::BN > n_valid;
#pragma unroll
            for (int j = 0; j < C::MJ; ++j) {
                float    bmin = 3.402823466e+38f;
                unsigned bidx = 0xffffffffu;
                const int tbase = t0 + wt * (C::BT / C::WT) + j * 32;
#pragma unroll
                for (int ih = 0; ih < 2; ++ih) {
                    float4 b4[2][4];
#pragma unroll
                    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq)
                            b4[i2][gq] = *(const float4*)(sb + ih * 64 + i2 * 32 + 8 * gq + 4 * hh);
#pragma unroll
                    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const f32x16& c  = acc[ih * 2 + i2][j];
                            const float4  bv = b4[i2][gq];
                            // score = -(activation + bias) = (-bias) - activation (exact)
                            const float4 v = make_float4((-bv.x) - c[gq * 4 + 0], (-bv.y) - c[gq * 4 + 1], (-bv.z) - c[gq * 4 + 2],
                                                         (-bv.w) - c[gq * 4 + 3]);
                            *(float4*)(w_lds + pipe_swz(tl32, i2 * 8 + 2 * gq + hh)) = v;
                            if (want_best) {
                                const int   n     = nbase + ih * 64 + i2 * 32 + 8 * gq + 4 * hh;
                                const float sc[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const bool ok = edge_n ? (n + e < n_valid) : true;
                                    if (ok && sc[e] < bmin) {  // ascending n within the lane
                                        bmin = sc[e];
                                        bidx = (unsigned)(n + e);
                                    }
                                }
                            }
                        }
                    // LDS -> global: 8 accesses of 4 frames x 256 B
                    float4 o[8];
#pragma unroll
                    for (int it = 0; it < 8; ++it)
                        o[it] = *(const float4*)(w_lds + pipe_swz(it * 4 + prow, pc));
                    const int n = nbase + ih * 64 + 4 * pc;
                    if ((DBG & 2) && o[0].x != 123.456f) {
                    }
                    else if (interior) {
                        // wave-uniform row base + one 32-bit lane offset (4 frames x 256 B per access)
                        const unsigned voff = (unsigned)(prow * ldo + 4 * pc) * 4u;
#pragma unroll
                        for (int it = 0; it < 8; ++it) {  // written once, never re-read here: keep the operand panels in L2
                            char* ub = (char*)((float*)out + (size_t)(tbase + it * 4) * ldo + nbase + ih * 64);
                            amx::nt_store(f32x4{o[it].x, o[it].y, o[it].z, o[it].w}, (f32x4*)(ub + voff));
                        }
                    }
                    else {
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const int t = tbase + it * 4 + prow;
                            if (t < t_valid) {
                                float* po = (float*)out + (size_t)t * ldo + n;
                                if (n + 3 < n_valid && out_aligned)
                                    *(float4*)po = o[it];
                                else {
                                    if (n < n_valid) po[0] = o[it].x;
                                    if (n + 1 < n_valid) po[1] = o[it].y;
                                    if (n + 2 < n_valid) po[2] = o[it].z;
                                    if (n + 3 < n_valid) po[3] = o[it].w;
                                }
                            }
                        }
                    }
                }
                if (want_best) {
                    // lanes l and l+32 hold the same frame, interleaved n: smaller index wins ties
                    const float    om = __shfl_xor(bmin, 32, 64);
                    const unsigned oi = (unsigned)__shfl_xor((int)bidx, 32, 64);
                    if (om < bmin || (om == bmin && oi < bidx)) {
                        bmin = om;
                        bidx = oi;
                    }
                    if (elane < 32) {
                        const int tl = wt * (C::BT / C::WT) + j * 32 + tl32;
                        s_min[wn * C::BT + tl] = bmin;
                        s_idx[wn * C::BT + tl] = bidx;
                    }
                }
            }
            if (want_best) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // s_min / s_idx written; no VMEM drain here
                __builtin_amdgcn_s_barrier();
                for (int tl = wave * 64 + elane; tl < C::BT; tl += C::THREADS) {
                    float    bmin = s_min[tl];
                    unsigned bidx = s_idx[tl];
#pragma unroll
                    for (int w = 1; w < C::WN; ++w) {  // ascending n ranges: strict '<' keeps the first minimum
                        const float m = s_min[w * C::BT + tl];
                        if (m < bmin) {
                            bmin = m;
                            bidx = s_idx[w * C::BT + tl];
                        }
                    }
                    part_min[(size_t)cur_tile_n * part_ld + t0 + tl] = bmin;
                    part_idx[(size_t)cur_tile_n * part_ld + t0 + tl] = bidx;
                }
            }
        }
        else {
#pragma unroll
            for (int j = 0; j < C::MJ; ++j) {
                const int tbase = t0 + wt * (C::BT / C::WT) + j * 32;
                uint2     lo[C::X3 ? C::MI : 1][4];  // split bf16: the lo plane of the 32 frames x 128 units, stored in a second pass
#pragma unroll
                for (int i = 0; i < C::MI; ++i) {
                    float4 b4[4];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        b4[gq] = *(const float4*)(sb + i * 32 + 8 * gq + 4 * hh);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x16& c  = acc[i][j];
                        const float   a0 = activate<ACT>(c[gq * 4 + 0] + b4[gq].x), a1 = activate<ACT>(c[gq * 4 + 1] + b4[gq].y);
                        const float   a2 = activate<ACT>(c[gq * 4 + 2] + b4[gq].z), a3 = activate<ACT>(c[gq * 4 + 3] + b4[gq].w);
                        uint2         pk;
                        pk.x = pack_bf16(a0, a1);
                        pk.y = pack_bf16(a2, a3);
                        if (C::X3) {  // lo = bf16(v - hi)
                            lo[i][gq].x = pack_bf16(a0 - __uint_as_float(pk.x << 16), a1 - __uint_as_float(pk.x & 0xffff0000u));
                            lo[i][gq].y = pack_bf16(a2 - __uint_as_float(pk.y << 16), a3 - __uint_as_float(pk.y & 0xffff0000u));
                        }
                        // the 8-byte half inside the chunk alternates with bit 3 of the row (b64 writes of rows r, r+8)
                        *(uint2*)(w_lds + pipe_swz(tl32, i * 4 + gq) + 8 * (hh ^ ((tl32 >> 3) & 1))) = pk;
                    }
                }
#pragma unroll
                for (int pl = 0; pl < C::PLANES; ++pl) {
                    if (pl == 1) {
#pragma unroll
                        for (int i = 0; i < C::MI; ++i)
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq)
                                *(uint2*)(w_lds + pipe_swz(tl32, i * 4 + gq) + 8 * (hh ^ ((tl32 >> 3) & 1))) = lo[C::X3 ? i : 0][gq];
                    }
                    uint4 o[8];
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const uint4 v = *(const uint4*)(w_lds + pipe_swz(it * 4 + prow, pc));
                        o[it]         = ((it >> 1) & 1) ? make_uint4(v.z, v.w, v.x, v.y) : v;  // rows 8-15, 24-31: halves swapped
                    }
                    const unsigned voff = (unsigned)(prow * ldo + 8 * pc) * 2u;  // wave-uniform row base + one 32-bit lane offset
#pragma unroll
                    for (int it = 0; it < 8; ++it)  // padded activation buffer: no guards
                        *(uint4*)((char*)((bf16_t*)out + (size_t)(tbase + it * 4) * ldo + pl * ld.olo + nbase) + voff) = o[it];
                }
            }
        }
        counted = interior && !(DBG & 2);
        if ((DBG & 1) && tid == 0)
            g_gemm_trace[((size_t)blockIdx.x * 64 + (step & 63)) * 4 + 2] = wall_clock64();
        if (!has_next)
            break;
        vi = nvi;
        bias_buf ^= 1;
    }
}

// combines the per-tile arg-min partials: best state per frame, per-state counts, sum of best scores.
// A workgroup owns 64 frames; its 4 waves take the tiles k = w, w + 4, ... (the GMM scorer has 625 of them) and meet in LDS.
// "First minimum in ascending state order" = smallest (value, state index) pair, so the order of combination is free.
__global__ __launch_bounds__(256) void best_state_reduce_kernel(const float* __restrict__ part_min, const unsigned* __restrict__ part_idx,
                                                               int n_tiles_n, int part_ld, int T, unsigned* __restrict__ best_state,
                                                               unsigned long long* __restrict__ counts, double* __restrict__ score_sum) {
    __shared__ float    s_min[4][64];
    __shared__ unsigned s_idx[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t    = blockIdx.x * 64 + lane;
    {
        float    bmin = 3.402823466e+38f;
        unsigned bidx = 0xffffffffu;
        if (t < T)
            for (int k = wave; k < n_tiles_n; k += 4) {
                const float    m = part_min[(size_t)k * part_ld + t];
                const unsigned i = part_idx[(size_t)k * part_ld + t];
                if (m < bmin || (m == bmin && i < bidx)) {
                    bmin = m;
                    bidx = i;
                }
            }
        s_min[wave][lane] = bmin;
        s_idx[wave][lane] = bidx;
    }
    __syncthreads();
    if (wave != 0)
        return;
    double   sum = 0.0;
    unsigned my  = 0xffffffffu;
    if (t < T) {
        float    bmin = s_min[0][lane];
        unsigned bidx = s_idx[0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float    m = s_min[w][lane];
            const unsigned i = s_idx[w][lane];
            if (m < bmin || (m == bmin && i < bidx)) {
                bmin = m;
                bidx = i;
            }
        }
        if (best_state)
            best_state[t] = bidx;
        if (bidx != 0xffffffffu)
            sum = (double)bmin;
        my = bidx;
    }
    // per-state counts: lanes of a wave that chose the same state issue ONE atomic (neighbouring frames
    // often share the best state, and a hot counter would otherwise serialise 32768 atomics per pass)
    {
        unsigned long long todo = __ballot(my != 0xffffffffu);
        while (todo) {
            const int      leader = __ffsll((long long)todo) - 1;
            const unsigned key    = (unsigned)__shfl((int)my, leader, 64);
            const unsigned long long same = __ballot(my == key) & todo;
            if (lane == leader)
                atomicAdd(&counts[key], (unsigned long long)__popcll(same));
            todo &= ~same;
        }
    }
    // wave-level sum, one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        sum += __shfl_xor(sum, off, 64);
    if (lane == 0 && sum != 0.0)
        atomicAdd(score_sum, sum);
}

// ---------------------------------------------------------------------------------------------
// fp32 GEMM (parity mode): exact f32 MFMA, register-staged LDS tiles with padded rows.
constexpr int BN = 128, BT = 128;  // f32 tile
constexpr int FK  = 32;       // k per tile
constexpr int FLD = FK + 1;   // padded row (floats) -> conflict-free column reads

template<int ACT, bool LAST>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ W, const float* __restrict__ X,
                                                      const float* __restrict__ bias, float* __restrict__ out, int Kpad, int ldx, int ldo,
                                                      int n_valid, int t_valid, int n_tiles_n) {
    __shared__ float sW[BN * FLD];
    __shared__ float sX[BT * FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wt = wave & 1;
    const int tile_t = blockIdx.x / n_tiles_n, tile_n = blockIdx.x - tile_t * n_tiles_n;
    const int n0 = tile_n * BN, t0 = tile_t * BT;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;

    for (int k0 = 0; k0 < Kpad; k0 += FK) {
        // 128 rows x 32 floats per operand = 1024 float4, 4 per thread
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int    idx = i * 256 + tid;
            const int    row = idx >> 3, c4 = (idx & 7) * 4;
            const float4 w   = *(const float4*)(W + (size_t)(n0 + row) * Kpad + k0 + c4);
            const float4 x   = *(const float4*)(X + (size_t)(t0 + row) * ldx + k0 + c4);
            float*       dw  = sW + row * FLD + c4;
            float*       dx  = sX + row * FLD + c4;
            dw[0] = w.x; dw[1] = w.y; dw[2] = w.z; dw[3] = w.w;
            dx[0] = x.x; dx[1] = x.y; dx[2] = x.z; dx[3] = x.w;
        }
        __syncthreads();
        const int frow = lane & 31, fk = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < FK / 2; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = sW[(wn * 64 + i * 32 + frow) * FLD + 2 * kk + fk];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[j] = sX[(wt * 64 + j * 32 + frow) * FLD + 2 * kk + fk];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = t0 + wt * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][g * 4 + e] + bias[n + e];
                    if (LAST) {
                        if (t < t_valid && n + e < n_valid)
                            out[(size_t)t * ldo + n + e] = -v;
                    }
                    else
                        out[(size_t)t * ldo + n + e] = activate<ACT>(v);
                }
            }
        }
}

}  // namespace amx

// ------------------------------------------------------------------------------------ ABI

struct amx_ffnn {
    amx_ctx*           ctx = nullptr;
    int                n_layers = 0, precision = AMX_PREC_BF16;
    std::vector<int>   in, out, act, Kpad, Npad;
    std::vector<void*> d_W;      // per layer [Npad x Kpad] bf16 or f32
    std::vector<float*> d_bias;  // per layer [Npad]; output layer has -alpha*logprior folded in
    // workspace (grown on demand)
    int    cap_T = 0;
    void*  d_in  = nullptr;      // packed input [cap_T x Kpad0]
    void*  d_act[2] = {nullptr, nullptr};
    std::vector<float> h_Wout, h_bout;  // output layer [n_emissions x K] f32 and its folded bias (on-demand scorer; uploaded on first use)
    bool   class_mapped = false; // a class-label wrapper reordered the output layer (amx_ffnn_forward_dev wants the network's own order)
    float* d_rowstat = nullptr;  // softmax top layer: per-frame maximum / sum [2][cap]
    int    rowstat_cap = 0;
    float *d_Wout = nullptr, *d_bout = nullptr;
    int    max_hidden_pad = 0;
    int    largest_layer  = 0;
    int    group_t = -1, group_n = -1;  // super-tile of the XCD-aware tile order
    // fused best-state statistics (amx_ffnn_score_stats_dev): per n-tile arg-min partials [ntn x Tpad]
    float*    d_part_min = nullptr;
    unsigned* d_part_idx = nullptr;
    size_t    part_cap   = 0;
    float*    cur_part_min = nullptr;
    unsigned* cur_part_idx = nullptr;
    int       cur_ntn      = 0;
    // HIP graphs of whole forward passes, for small batches where the 8-10 launches of a pass cost as much as a third of it
    struct GraphKey {
        const void *feats, *scores, *best, *counts, *sum;
        hipStream_t stream;
        int         stride, T, stats;
        bool operator<(const GraphKey& o) const {
            return std::tie(feats, scores, best, counts, sum, stream, stride, T, stats) <
                   std::tie(o.feats, o.scores, o.best, o.counts, o.sum, o.stream, o.stride, o.T, o.stats);
        }
    };
    std::map<GraphKey, hipGraphExec_t> graphs;
    float *d_host_f = nullptr, *d_host_s = nullptr;  // staging buffers of the host-buffer entry point amx_ffnn_score
    size_t host_f_cap = 0, host_s_cap = 0;
    int    use_graphs = 1;
    int    gemm_persistent = 1;
    int    gemm_cfg       = -1;  // -1 = automatic; index into the bf16 tile configurations (launch_bf16_cfg); tuning "tile"
    int    chunk          = 32768;  // frames per internal pass (tuning "chunk")
    int    mx_stagger     = 0;   // 10 ns ticks per XCD of gemm_mx_kernel's staggered start (0 = off, the default; tuning "stagger")
    int    mx_dbg         = 0;   // lab builds: ablation variant of gemm_mx_kernel (tuning "mx_dbg")
    int    mx_ksplit      = 1;   // tuning "ksplit": 4 = small batches (the one-tile-per-CU configuration with fewer tiles than a quarter of the
                                 // CUs) split K over four workgroups per tile (gemm_mx_kernel's comment)
    float*    d_ks_ws  = nullptr;   // split-K workspace: partial sums [tile][group][wave][register][lane]
    size_t    ks_ws_cap = 0;
    // AMX_PREC_F16MX: host-mapped word the kernels set when a value leaves the f16 range (sticky: every later call fails)
    unsigned* h_overflow = nullptr;
    unsigned* d_overflow = nullptr;
    size_t elt() const { return precision == AMX_PREC_FP32 ? 4 : 2; }
    bool   mfma_bf16() const { return precision != AMX_PREC_FP32; }  // everything but the exact-f32 kernels (fused statistics, HIP graphs)
    bool   is_mx() const { return precision == AMX_PREC_F16MX; }
    int    requested_precision = 0;   // amx_ffnn_model.precision; `precision` is what the handle computes in (mx_fallback)
    double mx_block_ratio      = 0.0; // AMX_PREC_F16MX requested: largest rms(block maxima) / rms(elements) over the layers
#ifdef AMX_LAB  // an ablation variant computes on stale operands: its scores are never valid, the flag is not looked at
    int    overflowed() const { return mx_dbg == 0 && h_overflow && *(volatile unsigned*)h_overflow; }
#else
    int    overflowed() const { return h_overflow && *(volatile unsigned*)h_overflow; }
#endif
};

namespace {

int pad_to(int v, int m) {
    return (v + m - 1) / m * m;
}

int ensure_workspace(amx_ffnn* h, int Tpad) {
    if (Tpad <= h->cap_T)
        return AMX_OK;
    // captured passes hold the old workspace addresses: drop them before the buffers move
    for (auto& kv : h->graphs)
        if (kv.second)
            hipGraphExecDestroy(kv.second);
    h->graphs.clear();
    hipFree(h->d_in);
    hipFree(h->d_act[0]);
    hipFree(h->d_act[1]);
    h->d_in = h->d_act[0] = h->d_act[1] = nullptr;
    h->cap_T                            = 0;
    const size_t planes = h->precision == AMX_PREC_BF16X3 ? 2 : 1;  // split bf16: rows are [hi plane | lo plane]
    if (h->is_mx()) {  // 25 KB blocks per (256 rows, 32 k)
        // The split-K workspace (tuning ksplit) is sized HERE, with the other buffers and outside any stream capture -- launch_mx
        // used to grow it lazily, which on a retry after a failed allocation happened inside hipStreamBeginCapture and wrote through an
        // iterator of the graph map it had just cleared (advisor, round 5).  Upper bound over the layers that can run split at this
        // batch size: tiles of 128 x 64, four computing waves of two 32 x 32 blocks, 16 x 64 floats each.  A failed allocation
        // switches the split off for the life of the handle (the default order: bit-identical to a handle created without ksplit).
        if (h->mx_ksplit > 1) {
            size_t need = 0;
            for (int l = 0; l < h->n_layers; ++l) {
                const long tiles = (long)(h->Npad[l] / 128) * (Tpad / 64);
                if (tiles * h->mx_ksplit <= (long)std::max(h->ctx->n_cu, 8) && h->Kpad[l] / 32 >= 8 * h->mx_ksplit)
                    need = std::max(need, (size_t)tiles * h->mx_ksplit * 4 * 2 * 16 * 64);
            }
            if (need > h->ks_ws_cap) {
                hipFree(h->d_ks_ws);
                h->d_ks_ws   = nullptr;
                h->ks_ws_cap = 0;
                if (hipMalloc((void**)&h->d_ks_ws, need * 4) != hipSuccess) {
                    (void)hipGetLastError();
                    h->mx_ksplit = 1;
                }
                else
                    h->ks_ws_cap = need;
            }
        }
        AMX_HIP(hipMalloc(&h->d_in, (size_t)(Tpad / 256) * (h->Kpad[0] / 32) * amx::mx::BLK));
        if (h->max_hidden_pad > 0) {
            AMX_HIP(hipMalloc(&h->d_act[0], (size_t)(Tpad / 256) * (h->max_hidden_pad / 32) * amx::mx::BLK));
            AMX_HIP(hipMalloc(&h->d_act[1], (size_t)(Tpad / 256) * (h->max_hidden_pad / 32) * amx::mx::BLK));
        }
        h->cap_T = Tpad;
        return AMX_OK;
    }
    AMX_HIP(hipMalloc(&h->d_in, (size_t)Tpad * planes * h->Kpad[0] * h->elt()));
    if (h->max_hidden_pad > 0) {
        AMX_HIP(hipMalloc(&h->d_act[0], (size_t)Tpad * planes * h->max_hidden_pad * h->elt()));
        AMX_HIP(hipMalloc(&h->d_act[1], (size_t)Tpad * planes * h->max_hidden_pad * h->elt()));
    }
    h->cap_T = Tpad;
    return AMX_OK;
}

// tile configurations of the bf16 GEMM, selected at run time (amx_ffnn_model.tuning tile=N overrides for experiments)
using CfgA = amx::GemmCfg<128, 128, 2, 2, 2>;  //  64 KB LDS, 2 workgroups per CU
using CfgC = amx::GemmCfg<256, 256, 2, 4, 2>;  // 128 KB LDS, 8 waves, wave tile 128x64
using CfgS = amx::GemmCfg<128, 64, 2, 2, 2>;   //  48 KB LDS, 3 workgroups per CU: small batches (fills the CUs)
using CfgS3L = amx::GemmCfg<128, 64, 2, 2, 3, 64, false, 4>;  // round 6: CfgS3 + four loader waves (hidden layers; opt-in tile=11: no gain measured)
using CfgS3 = amx::GemmCfg<128, 64, 2, 2, 3>;  //  72 KB LDS, two K-tiles in flight: layers with about one tile per CU (batch 1024 hidden layers: 19 -> 15.5 us)
// measured and dropped: GemmCfg<256,256,2,4,4,32> (4 stages of BK=32, three K-tiles in flight): 800 TF
// measured and dropped: 256x128x64 3-stage (753 TF), 256x256 with 64x128 wave tiles (973 TF) vs CfgC (1000 TF), CfgA (870 TF)
// split bf16: the same tile shapes with four planes (W_hi, W_lo, X_hi, X_lo) of 32 k per stage -- the same LDS bytes per stage as
// the 64-k bf16 tiles, 1.5 times the MFMA work per staged byte
using XfgA  = amx::GemmCfg<128, 128, 2, 2, 2, 32, true>;
using XfgC  = amx::GemmCfg<256, 256, 2, 4, 2, 32, true>;
using XfgS  = amx::GemmCfg<128, 64, 2, 2, 2, 32, true>;
using XfgS3 = amx::GemmCfg<128, 64, 2, 2, 3, 32, true>;

// strides and plane offsets of layer l's launch (x: ldx elements per row, lo plane at xlo; out: ldo, olo)
amx::GemmLd gemm_ld(const amx_ffnn* h, int l, int ldx, int ldo) {
    amx::GemmLd ld;
    const bool  x3 = h->precision == AMX_PREC_BF16X3;
    ld.ldw = x3 ? 2 * h->Kpad[l] : h->Kpad[l];
    ld.wlo = x3 ? h->Kpad[l] : 0;
    ld.ldx = ldx;
    ld.xlo = x3 ? ldx / 2 : 0;
    ld.ldo = ldo;
    ld.olo = x3 ? ldo / 2 : 0;  // hidden layers only; the output layer writes f32 scores
    return ld;
}

template<class C, int ACT, bool LAST, int VAR>
void launch_bf16v(amx_ffnn* h, int l, const void* x, int ldx, void* out, int ldo, int T, int Tpad) {
    const int ntn = h->Npad[l] / C::BN, ntt = Tpad / C::BT;
    auto      k   = amx::gemm_bf16_kernel<C, ACT, LAST, VAR>;
    // super-tile order: the tiles one XCD holds at a time share operand panels in its L2 (tools/gemm_probe.hip: output
    // layer 1.66 -> 1.44 ms with 8x4 blocks of 256x256 tiles; row-major order re-fetches every W panel per tile)
    // small-batch configurations (BN = 128): all frame tiles of two weight panels per XCD, so a panel is read once (batch 1024: 8x2 0.188 ms, 2x4 0.202)
    const int gt = h->group_t >= 0 ? h->group_t : 8, gn = h->group_n >= 0 ? h->group_n : (C::BN == 128 ? 2 : 4);
    constexpr int lds_bytes = amx::gemm_scratch_bytes<C, LAST>() + C::BN * 4;
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int per_cu = std::max(1, (160 * 1024) / lds_bytes);
    int       grid   = std::min(ntn * ntt, per_cu * std::max(h->ctx->n_cu, 8));
    if (grid >= 8)
        grid &= ~7;  // keep blockIdx % 8 == tile index % 8 for every stride step
    if (h->gemm_persistent == 0)
        grid = ntn * ntt;
    hipLaunchKernelGGL(k, dim3(grid), dim3(C::THREADS), lds_bytes, h->ctx->stream, (const amx::bf16_t*)h->d_W[l],
                       (const amx::bf16_t*)x, h->d_bias[l], out, h->Kpad[l], gemm_ld(h, l, ldx, ldo), h->out[l], T, ntn, ntn * ntt, gt, gn,
                       LAST ? h->cur_part_min : nullptr, LAST ? h->cur_part_idx : nullptr, Tpad);
    if (LAST)
        h->cur_ntn = ntn;
}

template<class C, int ACT, bool LAST>
void launch_bf16(amx_ffnn* h, int l, const void* x, int ldx, void* out, int ldo, int T, int Tpad) {
    // VAR bits are experiment switches (tools/gemm_probe.hip); the library instantiates variant 0.
    launch_bf16v<C, ACT, LAST, 0>(h, l, x, ldx, out, ldo, T, Tpad);
}

// large batches: the cross-tile pipelined kernel
template<class C, int ACT, bool LAST>
void launch_bf16_pipe(amx_ffnn* h, int l, const void* x, int ldx, void* out, int ldo, int T, int Tpad) {
    using P       = amx::PipeLds<C, LAST>;
    const int ntn = h->Npad[l] / C::BN, ntt = Tpad / C::BT;
    auto      k   = amx::gemm_bf16_pipe_kernel<C, ACT, LAST>;
    const int gt = h->group_t >= 0 ? h->group_t : 16, gn = h->group_n >= 0 ? h->group_n : 8;  // bench: 16x8 / 32x8 / 8x8 within noise, 2x16 and row-major 2-10 % slower
    static_assert(P::BYTES <= 160 * 1024, "LDS budget");
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, P::BYTES);
    int grid = std::min(ntn * ntt, std::max(h->ctx->n_cu, 8));
    if (grid >= 8)
        grid &= ~7;  // keep blockIdx % 8 == tile index % 8 for every stride step
    const int aligned = LAST ? (((uintptr_t)out & 15) == 0 && (ldo & 3) == 0) : 1;
    hipLaunchKernelGGL(k, dim3(grid), dim3(C::THREADS), P::BYTES, h->ctx->stream, (const amx::bf16_t*)h->d_W[l], (const amx::bf16_t*)x,
                       h->d_bias[l], out, h->Kpad[l], gemm_ld(h, l, ldx, ldo), h->out[l], T, ntn, ntn * ntt, gt, gn, aligned,
                       LAST ? h->cur_part_min : nullptr, LAST ? h->cur_part_idx : nullptr, Tpad);
    if (LAST)
        h->cur_ntn = ntn;
}

// workgroups of the pipelined kernel for `tiles` 256 x 256 tiles (launch_bf16_pipe: one per CU, a multiple of 8 from 8 on)
static inline long pipe_grid(long tiles, long ncu) {
    long g = std::min(tiles, std::max(ncu, 8L));
    if (g >= 8)
        g &= ~7L;
    return std::max(g, 1L);
}

template<int ACT, bool LAST>
void launch_bf16_cfg(amx_ffnn* h, int l, const void* x, int ldx, void* out, int ldo, int T, int Tpad) {
    // default: 256x256 tiles when they still give >= 2 tiles per CU, else 128x128 (small batches)
    int cfg = h->gemm_cfg;
    if (cfg < 0) {
        const long ncu = std::max(h->ctx->n_cu, 1), t256 = (long)(h->Npad[l] / 256) * (Tpad / 256), t128 = (long)(h->Npad[l] / 128) * (Tpad / 128);
        if (t256 >= 2L * ncu)
            cfg = 2;
        // between one half and two tiles of 256 x 256 per CU (the output layer at batch 1024: 160 tiles): the pipelined kernel in ONE
        // round against the 128 x 128 tiles in ceil(tiles / 2 per CU) rounds -- a 256 x 256 tile takes 1.8 rounds of the small ones
        // (57 vs 32 us at K = 2048); output layer at batch 1024: 64 -> 57 us bf16, 147 -> 135 us split bf16
        // (rounds with the grid the launch really uses: launch_bf16_pipe rounds the workgroup count down to a multiple of 8, so
        // e.g. 129 tiles on 256 CUs run as 128 workgroups and one of them takes a second tile)
        else if (2 * t256 >= ncu && ((t256 + pipe_grid(t256, ncu) - 1) / pipe_grid(t256, ncu)) * 9 < ((t128 + 2 * ncu - 1) / (2 * ncu)) * 5)
            cfg = 2;
        else if ((long)(h->Npad[l] / 128) * (Tpad / 128) >= (long)h->ctx->n_cu)
            cfg = 0;
        else if ((long)(h->Npad[l] / 128) * (Tpad / 64) <= 2L * h->ctx->n_cu)
            cfg = 6;  // about one tile per CU: nothing else hides the operand latency, keep two K-tiles in flight
        else
            cfg = 3;
    }
    if (h->precision == AMX_PREC_BF16X3) {
        switch (cfg) {
            case 2: launch_bf16_pipe<XfgC, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
            case 4: launch_bf16<XfgC, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
            case 3: launch_bf16<XfgS, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
            case 6: launch_bf16<XfgS3, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
            default: launch_bf16<XfgA, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
        }
        return;
    }
    switch (cfg) {
        case 2: launch_bf16_pipe<CfgC, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
        case 4: launch_bf16<CfgC, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;  // previous large-batch kernel (A/B runs)
        case 3: launch_bf16<CfgS, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
        case 6: launch_bf16<CfgS3, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
        case 11:   // round 6, A/B runs: CfgS3 with four loader waves for the hidden layers -- bit-identical, and measured NO gain (17.4 against
                   // 17.0 us per 2048 x 2048 layer at batch 1024): the tile is bound by the LDS time of its fragment reads (48 KB per K-tile)
            if constexpr (LAST)
                launch_bf16<CfgS3, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad);
            else
                launch_bf16<CfgS3L, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad);
            break;
        default: launch_bf16<CfgA, ACT, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
    }
}

// AMX_PREC_F16MX tile configurations (ffnn_mx.hpp): the same shapes and the same selection rule as the bf16 kernels
using MfgL = amx::mx::MxCfg<256, 256, 2, 4, 3>;      // 147 KB LDS, 8 waves of 128 x 64, two K-tiles in flight
using MfgLP = amx::mx::MxCfg<256, 256, 2, 4, 3, 4>;  // the same with the software L2 prefetch 4 K-tiles ahead (tuning tile=4: A/B runs; measured 6 % SLOWER: the
                                                     // prefetch loads sit in the in-order vmcnt queue in front of the next K-tile's pieces)
using MfgLH = amx::mx::MxCfg<256, 256, 2, 4, 3, 0, 4>;  // the first wave of every SIMD issues all LDS-DMA pieces (tuning tile=5: A/B runs)
using MfgLF = amx::mx::MxCfg<256, 256, 2, 4, 3, 4, 4>;  // ... and its partner prefetches four K-tiles ahead into L2, outside every counted queue (tile=7)
using MfgLQ = amx::mx::MxCfg<256, 256, 2, 4, 3, 0, 0, 1, 0, 1>;  // round 5: ping-pong halves -- one wave of a SIMD issues its products while its partner reads / refills (tile=8)
using MfgLW = amx::mx::MxCfg<256, 256, 2, 2, 3, 0, 0, 1, 0, 2>;  // round 5: ONE wave per SIMD, four waves of 128 x 128 that pipeline themselves (tile=9)
using MfgA = amx::mx::MxCfg<128, 128, 2, 2, 3>;  //  74 KB: 2 workgroups per CU
using MfgS = amx::mx::MxCfg<128, 64, 2, 2, 8, 0, 0, 2>;  // 147 KB: small batches, ONE tile per CU, two K-tiles per barrier, four more in flight (a
                                                         // 2048 x 2048 layer at batch 1024 is 64 K-tiles of 9 matrix instructions per wave: barrier and LDS round trip per K-tile were its time)
using MfgS4 = amx::mx::MxCfg<128, 64, 2, 2, 4>;          //  74 KB, one K-tile per barrier, three in flight (tuning tile=6: A/B runs)
using MfgSL = amx::mx::MxCfg<128, 64, 2, 2, 8, 0, 0, 2, 4>;  // hidden layers: MfgS + four loader waves (512 threads: a computing and a loading wave per SIMD)
using MfgS8 = amx::mx::MxCfg<128, 64, 4, 2, 8, 0, 0, 2>;  // round 6 (tile=12): the one-tile-per-CU tile on EIGHT computing waves of 32 x 32 (two per SIMD: one wave's reads and conversions under the other's products), every wave issuing its share of the DMA
using Mfg64 = amx::mx::MxCfg<64, 64, 2, 2, 8, 0, 0, 4, 4>;    // round 6 (tile=14): 64 x 64 tiles with loader waves, FOUR K-tiles per barrier (96 KB ring): hidden layers of fills that leave half the CUs without a 128 x 64 tile
using MfgSP = amx::mx::MxCfg<128, 64, 2, 2, 8, 0, 0, 2, 4, 3>;  // round 6, the default for hidden layers of small batches: MfgSL with the READ-AHEAD K loop (two register images, conversions issued first, a steady-state body without run-time wait selection); MfgSL stays as tile=11

template<class C, int ACT, bool LAST>
void launch_mx(amx_ffnn* h, int l, const void* x, int xkts, void* out, int ldo, int T, int Tpad, int n_valid) {
    const int ntn = h->Npad[l] / C::BN, ntt = Tpad / C::BT;
    const int gt = h->group_t >= 0 ? h->group_t : (C::BN == 256 ? 16 : 8), gn = h->group_n >= 0 ? h->group_n : (C::BN == 256 ? 8 : 2);
    constexpr int lds_bytes = amx::gemm_scratch_bytes<C, LAST>() + C::BN * 4;   // stages / epilogue scratch + the tile's bias
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    const int per_cu = std::max(1, (160 * 1024) / lds_bytes);
    // split-K across workgroups (tuning ksplit=4): only where it multiplies the CUs that pull operands -- the one-tile-per-CU
    // configuration with at most a quarter of the CUs busy and enough K-tiles per group to fill the ring
    int ksplit = 1;
    if (h->mx_ksplit > 1 && C::BN == 128 && C::BT == 64 && C::U == 2 && (long)ntn * ntt * h->mx_ksplit <= (long)std::max(h->ctx->n_cu, 8) &&
        h->Kpad[l] / 32 >= 8 * h->mx_ksplit)
        ksplit = h->mx_ksplit;
    if (ksplit > 1 && (size_t)ntn * ntt * ksplit * C::NW * C::MI * C::MJ * 16 * 64 > h->ks_ws_cap)
        ksplit = 1;  // (ensure_workspace sizes the workspace for every layer that can run split; nothing is allocated here: this may be inside a stream capture)
    int       grid   = std::min(ntn * ntt * ksplit, per_cu * std::max(h->ctx->n_cu, 8));
    if (grid >= 8)
        grid &= ~7;  // keep blockIdx % 8 == tile index % 8 for every stride step
    // staggered XCDs (see the kernel): off by default -- 0 / 1 / 3 / 6 us per XCD measured within noise of one another
    // (profiles/r04/stagger.log: output layer 2.085-2.111 ms in every setting)
    const int stagger = (LAST && C::BN == 256 && ntn * ntt >= 8 * grid && h->mx_stagger > 0) ? h->mx_stagger : 0;
#define AMX_MX_LAUNCH(DBG)                                                                                                                  \
    do {                                                                                                                                    \
        auto k = amx::mx::gemm_mx_kernel<C, ACT, LAST, DBG>;                                                                                \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);                                         \
        hipLaunchKernelGGL(k, dim3(grid), dim3(C::THREADS), lds_bytes, h->ctx->stream, (const char*)h->d_W[l], (const char*)x, h->d_bias[l], \
                           out, h->Kpad[l] / 32, xkts, h->Npad[l] / 32, ldo, n_valid, T, ntn, ntn * ntt, gt, gn,                            \
                           LAST ? h->cur_part_min : nullptr, LAST ? h->cur_part_idx : nullptr, Tpad, h->d_overflow, stagger, ksplit,       \
                           h->d_ks_ws);                                                                                                     \
        if (ksplit > 1) /* launch 2: the partial sums of every tile, in group order, then the epilogue */                                  \
            hipLaunchKernelGGL(k, dim3(std::min(ntn * ntt, grid)), dim3(C::THREADS), lds_bytes, h->ctx->stream, (const char*)h->d_W[l],     \
                               (const char*)x, h->d_bias[l], out, h->Kpad[l] / 32, xkts, h->Npad[l] / 32, ldo, n_valid, T, ntn, ntn * ntt,  \
                               gt, gn, LAST ? h->cur_part_min : nullptr, LAST ? h->cur_part_idx : nullptr, Tpad, h->d_overflow, 0, -ksplit, \
                               h->d_ks_ws);                                                                                                 \
    } while (0)
    int dbg = 0;
#ifdef AMX_LAB  // ablations of the large-batch kernel (tools/ab_mx.sh, profiles/r04/gemm_mx_ablation.log)
    dbg = h->mx_dbg;
    if constexpr (C::BN == 256 && C::PIPE == 0 && ACT == AMX_ACT_RELU * (LAST ? 0 : 1)) {
        switch (dbg) {
            case 8: AMX_MX_LAUNCH(8); break;
            case 24: AMX_MX_LAUNCH(24); break;
            case 32: AMX_MX_LAUNCH(32); break;
            case 64: AMX_MX_LAUNCH(64); break;
            case 72: AMX_MX_LAUNCH(72); break;
            case 128: AMX_MX_LAUNCH(128); break;
            case 136: AMX_MX_LAUNCH(136); break;
            case 152: AMX_MX_LAUNCH(152); break;
            case 256: AMX_MX_LAUNCH(256); break;
            case 280: AMX_MX_LAUNCH(280); break;
            case 512: AMX_MX_LAUNCH(512); break;
            case 768: AMX_MX_LAUNCH(768); break;
            case 192: AMX_MX_LAUNCH(192); break;
            case 1024: AMX_MX_LAUNCH(1024); break;
            case 2048: AMX_MX_LAUNCH(2048); break;
            case 2304: AMX_MX_LAUNCH(2304); break;
            case 2560: AMX_MX_LAUNCH(2560); break;
            case 2816: AMX_MX_LAUNCH(2816); break;
            case 4096: AMX_MX_LAUNCH(4096); break;
            case 6144: AMX_MX_LAUNCH(6144); break;
            case 1032: AMX_MX_LAUNCH(1032); break;
            case 1160: AMX_MX_LAUNCH(1160); break;
            case 16: AMX_MX_LAUNCH(16); break;
            default: dbg = 0; break;
        }
    }
    else if (C::BN == 128 && C::BT == 64 && C::PIPE == 3 && !LAST && ACT == AMX_ACT_RELU && (dbg == 4 || dbg == 8 || dbg == 16 || dbg == 20 || dbg == 24 || dbg == 28 || dbg == 128 || dbg == 156)) {
        // round 6: the ablations of the one-tile-per-CU configuration with read-ahead (tools/hidden_layer_probe.py): 4 no fragment
        // reads, 8 no matrix instructions, 16 no operand DMA behind the prologue, and their sums
        if constexpr (C::BN == 128 && C::BT == 64 && C::PIPE == 3 && !LAST && ACT == AMX_ACT_RELU) {
            switch (dbg) {
                case 4: AMX_MX_LAUNCH(4); break;
                case 8: AMX_MX_LAUNCH(8); break;
                case 16: AMX_MX_LAUNCH(16); break;
                case 20: AMX_MX_LAUNCH(20); break;
                case 24: AMX_MX_LAUNCH(24); break;
                case 128: AMX_MX_LAUNCH(128); break;
                case 156: AMX_MX_LAUNCH(156); break;
                default: AMX_MX_LAUNCH(28); break;
            }
        }
    }
    else if (dbg == 2048 && ACT == AMX_ACT_RELU * (LAST ? 0 : 1)) {
        AMX_MX_LAUNCH(2048);  // time stamps of any tile configuration (tools/mx_timeline.py small)
    }
    else if (dbg == 128 && C::PIPE == 1 && ACT == AMX_ACT_RELU * (LAST ? 0 : 1)) {
        if constexpr (C::PIPE == 1)
            AMX_MX_LAUNCH(128);  // the ping-pong tile without conversions and lane swaps (timing only: what a q that travels with the operands would save)
    }
    else
        dbg = 0;
#endif
    if (dbg == 0)
        AMX_MX_LAUNCH(0);
#undef AMX_MX_LAUNCH
    if (LAST)
        h->cur_ntn = ntn;
}

template<int ACT, bool LAST>
void launch_mx_cfg(amx_ffnn* h, int l, const void* x, int xkts, void* out, int ldo, int T, int Tpad, int n_valid) {
    int cfg = h->gemm_cfg;
    if (cfg < 0) {
        const long ncu = std::max(h->ctx->n_cu, 1), t256 = (long)(h->Npad[l] / 256) * (Tpad / 256), t128 = (long)(h->Npad[l] / 128) * (Tpad / 128);
        const long t64 = (long)(h->Npad[l] / 128) * (Tpad / 64);  // tiles of 128 x 64
        if (t256 >= 2L * ncu)
            cfg = 8;  // round 5: the 256 x 256 tile with the ping-pong K loop (tile=2: all eight waves in phase, round 4's default)
        else if (LAST && t256 <= ncu && 5 * t256 >= 3 * ncu)
            // an output layer whose 256 x 256 tiles fill 60-100 % of the CUs in ONE round (config 4 at its own batch: 40 x 4 = 160 tiles):
            // 87.5 us against 95 for 632 tiles of 128 x 128 on 512 slots (a full round and a quarter-full one); bit-identical
            cfg = 8;
        else if (t128 >= ncu)
            cfg = 0;
        else if (t64 > ncu && t64 <= 2 * ncu)
            // between one and two tiles of 128 x 64 per CU (the output layer of a 256-frame fill: 316 tiles): the one-tile-per-CU
            // configuration would run a full round and a quarter-full second one; with the 74 KB ring two workgroups share a CU and
            // everything is resident at once -- output layer at 256 frames 87 -> 42 us (profiles/r05/fill_breakdown.log); the same
            // K order, the same matrix instructions: bit-identical
            cfg = 6;
        else if (!LAST && 2 * t64 <= ncu && h->mx_ksplit <= 1)  // (ksplit is the 128 x 64 tile's: a handle that asked for it keeps that tile)
            // at most half a tile of 128 x 64 per CU (a hidden layer of a 256- or 512-frame fill): 64 x 64 tiles, four K-tiles per
            // barrier -- twice the workgroups, half the iterations; 29.5 -> 27.7 us per 2048 x 2048 layer (hidden_layer_probe; two / three K-tiles
            // per barrier on the same tile: 29.3 / 28.3 us; 128 x 64 with four spills, 64 x 128 spills), the same
            // K order and matrix instructions per accumulator: bit-identical
            cfg = 14;
        else
            cfg = 3;
    }
    switch (cfg) {
        case 2: launch_mx<MfgL, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid); break;
        case 4: launch_mx<MfgLP, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid); break;
        case 5: launch_mx<MfgLH, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid); break;
        case 7: launch_mx<MfgLF, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid); break;
        case 8: launch_mx<MfgLQ, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid); break;
        case 9: launch_mx<MfgLW, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid); break;
        case 3:
            if constexpr (LAST)
                launch_mx<MfgS, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid);
            else
                launch_mx<MfgSP, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid);   // round 6: read-ahead K loop (bit-identical to MfgSL: tile=11)
            break;
        case 6: launch_mx<MfgS4, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid); break;
        case 12: launch_mx<MfgS8, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid); break;
        case 14:
            if constexpr (LAST)
                launch_mx<MfgS, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid);
            else
                launch_mx<Mfg64, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid);
            break;
case 11:   // round 5's hidden-layer loop (no read-ahead): A/B runs
            if constexpr (LAST)
                launch_mx<MfgS, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid);
            else
                launch_mx<MfgSL, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid);
            break;
        default: launch_mx<MfgA, ACT, LAST>(h, l, x, xkts, out, ldo, T, Tpad, n_valid); break;
    }
}

template<bool LAST>
int launch_layer(amx_ffnn* h, int l, const void* x, int ldx, void* out, int ldo, int T, int Tpad) {
    hipStream_t            st = h->ctx->stream;
    amx::ScopedKernelTimer t_all(h->ctx, "ffnn_gemm");
    hipEvent_t             e0 = nullptr, e1 = nullptr;
    const bool             time_max = h->ctx->profiling && l == h->largest_layer;
    if (time_max) {
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, st);
    }
    const int act = LAST ? AMX_ACT_NONE : h->act[l];
    if (h->is_mx()) {  // LAST with l < n_layers - 1: the last hidden layer as f32 rows (amx_ffnn_forward_hidden_dev)
        switch (act) {
            case AMX_ACT_RELU: launch_mx_cfg<AMX_ACT_RELU, LAST>(h, l, x, ldx, out, ldo, T, Tpad, h->out[l]); break;
            case AMX_ACT_SIGMOID: launch_mx_cfg<AMX_ACT_SIGMOID, LAST>(h, l, x, ldx, out, ldo, T, Tpad, h->out[l]); break;
            case AMX_ACT_TANH: launch_mx_cfg<AMX_ACT_TANH, LAST>(h, l, x, ldx, out, ldo, T, Tpad, h->out[l]); break;
            default: launch_mx_cfg<AMX_ACT_NONE, LAST>(h, l, x, ldx, out, ldo, T, Tpad, h->out[l]); break;
        }
    }
    else if (h->mfma_bf16()) {
        switch (act) {
            case AMX_ACT_RELU: launch_bf16_cfg<AMX_ACT_RELU, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
            case AMX_ACT_SIGMOID: launch_bf16_cfg<AMX_ACT_SIGMOID, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
            case AMX_ACT_TANH: launch_bf16_cfg<AMX_ACT_TANH, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
            default: launch_bf16_cfg<AMX_ACT_NONE, LAST>(h, l, x, ldx, out, ldo, T, Tpad); break;
        }
    }
    else {
        const int ntn = h->Npad[l] / amx::BN, ntt = Tpad / amx::BT;
        dim3      grid(ntn * ntt), block(256);
        const int nv = h->out[l];
#define AMX_L(ACT)                                                                                                   \
    hipLaunchKernelGGL((amx::gemm_f32_kernel<ACT, LAST>), grid, block, 0, st, (const float*)h->d_W[l], (const float*)x, \
                       h->d_bias[l], (float*)out, h->Kpad[l], ldx, ldo, nv, T, ntn);
        switch (act) {
            case AMX_ACT_RELU: AMX_L(AMX_ACT_RELU) break;
            case AMX_ACT_SIGMOID: AMX_L(AMX_ACT_SIGMOID) break;
            case AMX_ACT_TANH: AMX_L(AMX_ACT_TANH) break;
            default: AMX_L(AMX_ACT_NONE) break;
        }
#undef AMX_L
    }
    if (time_max) {
        hipEventRecord(e1, st);
        h->ctx->prof["ffnn_gemm_max"].events.emplace_back(e0, e1);
    }
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

}  // namespace

extern "C" {

int amx_ffnn_create(amx_ctx* ctx, const amx_ffnn_model* m, amx_ffnn** out) {
    AMX_REQUIRE(ctx && m && out, AMX_ERR_INVALID, "amx_ffnn_create: NULL argument");
    *out = nullptr;
    AMX_REQUIRE(m->n_layers >= 1 && m->in_dim && m->out_dim && m->W && m->bias && m->activation, AMX_ERR_INVALID,
                "amx_ffnn_create: empty network");
    AMX_REQUIRE(m->precision == AMX_PREC_FP32 || m->precision == AMX_PREC_BF16 || m->precision == AMX_PREC_BF16X3 ||
                        m->precision == AMX_PREC_F16MX,
                AMX_ERR_INVALID, "amx_ffnn_create: unknown precision");
    for (int l = 0; l < m->n_layers; ++l) {
        AMX_REQUIRE(m->in_dim[l] > 0 && m->out_dim[l] > 0 && m->W[l], AMX_ERR_INVALID, "amx_ffnn_create: layer %d is empty", l);
        if (l > 0)
            AMX_REQUIRE(m->in_dim[l] == m->out_dim[l - 1], AMX_ERR_INVALID,
                        "amx_ffnn_create: layer %d input dimension %d != previous output %d", l, m->in_dim[l], m->out_dim[l - 1]);
        AMX_REQUIRE(m->activation[l] >= AMX_ACT_NONE && m->activation[l] <= AMX_ACT_TANH, AMX_ERR_INVALID,
                    "amx_ffnn_create: unknown activation in layer %d", l);
    }
    // Nn::BatchFeatureScorer: "output layer must be of type 'linear+softmax'" with the softmax switched off
    AMX_REQUIRE(m->activation[m->n_layers - 1] == AMX_ACT_NONE, AMX_ERR_INVALID, "amx_ffnn_create: output layer must be linear (softmax is not evaluated)");

    // ---- Nn::ClassLabelWrapper (Nn/ClassLabelWrapper.cc:56-75, Nn/BatchFeatureScorer.cc:148-171): emission e reads network output
    // class_to_output[e]; a disregarded class (-1) scores Core::Type<f32>::max.  The mapping is one-to-one (the reference refuses
    // anything else), so it is applied ONCE, to the output layer: row e of the emission-ordered layer is row class_to_output[e]
    // of the network's (with its bias and prior), a disregarded class gets a zero row and bias -FLT_MAX -- score = -(0 + bias) =
    // FLT_MAX exactly, in every precision mode.  Scores, arg-min statistics and the on-demand scorer then index emissions.
    const int          L0 = m->n_layers - 1;
    std::vector<int>   out_dim(m->out_dim, m->out_dim + m->n_layers);
    std::vector<const float*> Wl(m->W, m->W + m->n_layers), bl(m->bias, m->bias + m->n_layers);
    const float*       prior = m->log_prior;
    std::vector<float> Wmap, bmap, pmap;
    const bool class_mapped = m->class_to_output != nullptr;
    if (m->class_to_output) {
        AMX_REQUIRE(m->n_classes > 0, AMX_ERR_INVALID, "amx_ffnn_create: class_to_output without n_classes");
        const int N = m->out_dim[L0], K = m->in_dim[L0];
        std::vector<char> used((size_t)N, 0);
        int               n_targets = 0;
        for (int e = 0; e < m->n_classes; ++e) {
            const int o = m->class_to_output[e];
            AMX_REQUIRE(o >= -1 && o < N, AMX_ERR_INVALID, "amx_ffnn_create: class %d maps to output %d (network has %d outputs)", e, o, N);
            if (o >= 0) {
                // ClassLabelWrapper::isOneToOneMapping: "no one-to-one correspondence between network outputs and classes!"
                AMX_REQUIRE(!used[o], AMX_ERR_INVALID, "amx_ffnn_create: no one-to-one correspondence between network outputs and classes (output %d)", o);
                used[o] = 1;
                ++n_targets;
            }
        }
        // require_eq(network_.getTopLayer().getOutputDimension(), labelWrapper_->nClassesToAccumulate())
        AMX_REQUIRE(n_targets == N, AMX_ERR_INVALID, "amx_ffnn_create: %d classes to accumulate, but the output layer has %d units", n_targets, N);
        Wmap.assign((size_t)m->n_classes * K, 0.f);
        bmap.assign((size_t)m->n_classes, -3.402823466e+38f);
        pmap.assign((size_t)m->n_classes, 0.f);
        for (int e = 0; e < m->n_classes; ++e) {
            const int o = m->class_to_output[e];
            if (o < 0)
                continue;
            memcpy(&Wmap[(size_t)e * K], m->W[L0] + (size_t)o * K, (size_t)K * 4);
            bmap[e] = m->bias[L0] ? m->bias[L0][o] : 0.f;
            pmap[e] = m->log_prior ? m->log_prior[o] : 0.f;
        }
        out_dim[L0] = m->n_classes;
        Wl[L0]      = Wmap.data();
        bl[L0]      = bmap.data();
        if (m->log_prior)
            prior = pmap.data();
    }

    amx::Tuning tune;
    {
        static const char* const keys[] = {"tile", "graph", "persistent", "group", "chunk", "stagger", "mx_dbg", "mx_fallback", "ksplit", nullptr};
        if (!tune.parse(m->tuning, keys, "amx_ffnn_create"))
            return AMX_ERR_INVALID;
    }
    // values are checked like keys (a typo must not silently select the default kernel)
    int t_tile, t_graph, t_persistent, t_chunk, t_mx_dbg, t_stagger, t_group_t = -1, t_group_n = -1, t_ksplit = 1;
    std::string t_mx_fallback;
    {
        const char* who = "amx_ffnn_create";
        static const char* const fallbacks[] = {"auto", "off", nullptr};
        if (!tune.get_word("mx_fallback", "auto", fallbacks, &t_mx_fallback, who) || !tune.get_int("ksplit", 1, 1, 4, &t_ksplit, who))
            return AMX_ERR_INVALID;
        AMX_REQUIRE(t_ksplit == 1 || t_ksplit == 4, AMX_ERR_INVALID, "amx_ffnn_create: tuning ksplit=%d: expected 1 | 4", t_ksplit);
        AMX_REQUIRE(t_ksplit == 1 || m->precision == AMX_PREC_F16MX, AMX_ERR_UNSUPPORTED, "amx_ffnn_create: tuning ksplit exists for AMX_PREC_F16MX only");
        if (!tune.get_int("tile", -1, -1, 14, &t_tile, who) || !tune.get_int("graph", 0, 0, 1, &t_graph, who) ||
            !tune.get_int("persistent", 1, 0, 1, &t_persistent, who) || !tune.get_int("chunk", 32768, 256, 1 << 24, &t_chunk, who) ||
            !tune.get_int("mx_dbg", 0, 0, 1 << 16, &t_mx_dbg, who) || !tune.get_int("stagger", 0, 0, 100000, &t_stagger, who))
            return AMX_ERR_INVALID;
        if (tune.has("group")) {
            const std::string g = tune.kv["group"];
            char              tail = 0;
            AMX_REQUIRE(sscanf(g.c_str(), "%dx%d%c", &t_group_t, &t_group_n, &tail) == 2 && t_group_t >= 1 && t_group_n >= 1 && t_group_t <= 4096 &&
                                t_group_n <= 4096,
                        AMX_ERR_INVALID, "amx_ffnn_create: tuning group=%s: expected <frame tiles>x<output tiles>, e.g. 16x8", g.c_str());
        }
    }
    // AMX_PREC_F16MX on heavy-tailed WEIGHTS.  One e8m0 exponent serves 32 k of a row pair; a block whose maximum dwarfs the rest leaves
    // the fp6 image q(w) of the small values at zero, and their cross terms r(x) q(w) are lost: measured (tests/test_ffnn_f16mx_gpu.py,
    // profiles/r05/f16mx_families.log) the error grows from ~27 x that of f32 accumulation (Gaussian weights) to 54-115 x (log-normal
    // rows, one outlier per block).  The statistic G = rms of the block maxima / rms of the elements says which case a layer is:
    // 2.4 Gaussian, 3.0 Laplace, 3.3 Student-t(4), 5.0 log-normal(1.5), 5.66 = sqrt(32) when one element carries every block.  Above
    // 4.0 the handle computes in split bf16 instead (tuning mx_fallback=auto, the default; mx_fallback=off keeps f16mx):
    // amx_ffnn_precision() reports what it runs.  Activations have no such check (they are not known here); an outlier per block in the
    // FEATURES costs 1.2 x (32 x against 27 x, unnormalised MFCC context windows included).
    int    prec = m->precision;
    double mx_ratio = 0.0;
    if (prec == AMX_PREC_F16MX) {
        for (int l = 0; l < m->n_layers; ++l) {
            const int    K = m->in_dim[l], N = out_dim[l];
            const float* W = Wl[l];
            double       sum_max2 = 0.0, sum_w2 = 0.0;
            long         n_blocks = 0;
            for (int n = 0; n < N; ++n)
                for (int k0 = 0; k0 < K; k0 += 32) {
                    float mx = 0.f;
                    for (int k = k0; k < std::min(K, k0 + 32); ++k) {
                        const float v = W[(size_t)n * K + k];
                        mx            = std::fmax(mx, std::fabs(v));
                        sum_w2 += (double)v * (double)v;
                    }
                    sum_max2 += (double)mx * (double)mx;
                    ++n_blocks;
                }
            if (sum_w2 > 0.0)
                mx_ratio = std::max(mx_ratio, std::sqrt((sum_max2 / (double)n_blocks) / (sum_w2 / ((double)N * (double)K))));
        }
        if (t_mx_fallback == "auto" && mx_ratio > 4.0) {
            // the handle computes in split bf16: twice the matrix work, other rounding than the caller asked for -- said ONCE, on
            // stderr (the adapter logs amx_ffnn_precision as well), and a split-K request -- an f16mx schedule -- is refused rather
            // than accepted and ignored (advisor, round 5)
            AMX_REQUIRE(t_ksplit <= 1, AMX_ERR_INVALID,
                        "amx_ffnn_create: tuning ksplit=%d applies to AMX_PREC_F16MX, but this network's weights are heavy-tailed (block-maximum "
                        "statistic %.2f > 4.0) and the handle would compute in AMX_PREC_BF16X3: drop ksplit, or keep f16mx with mx_fallback=off",
                        t_ksplit, mx_ratio);
            std::fprintf(stderr, "rasr_amd: amx_ffnn_create: AMX_PREC_F16MX requested, computing in AMX_PREC_BF16X3 (block-maximum statistic of the "
                                 "weights %.2f > 4.0; tuning mx_fallback=off keeps f16mx)\n", mx_ratio);
            prec = AMX_PREC_BF16X3;
        }
    }
    amx_ffnn* h  = new amx_ffnn;
    h->ctx       = ctx;
    h->n_layers  = m->n_layers;
    h->precision = prec;
    h->requested_precision = m->precision;
    h->mx_block_ratio      = mx_ratio;
    h->class_mapped = class_mapped;
    h->gemm_cfg        = t_tile;
    h->use_graphs      = t_graph;
    h->gemm_persistent = t_persistent;
    h->chunk           = t_chunk;
    h->mx_dbg          = t_mx_dbg;
    h->mx_stagger      = t_stagger;
    h->mx_ksplit       = t_ksplit;
    h->group_t         = t_group_t;
    h->group_n         = t_group_n;
    hipSetDevice(ctx->device);
    const int kmult = prec == AMX_PREC_F16MX ? amx::mx::TK : (prec != AMX_PREC_FP32) ? amx::BK : amx::FK;
    if (prec == AMX_PREC_F16MX) {
        if (hipHostMalloc((void**)&h->h_overflow, 4, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&h->d_overflow, h->h_overflow, 0) != hipSuccess) {
            (void)hipGetLastError();
            amx::set_error("amx_ffnn_create: cannot allocate the overflow flag");
            amx_ffnn_destroy(h);
            return AMX_ERR_DEVICE;
        }
        *h->h_overflow = 0;
    }
    long      best_flops = -1;
    for (int l = 0; l < m->n_layers; ++l) {
        h->in.push_back(m->in_dim[l]);
        h->out.push_back(out_dim[l]);
        h->act.push_back(m->activation[l]);
        // hidden activations are stored with a row stride of Npad(l-1) >= Kpad(l)
        h->Kpad.push_back(pad_to(m->in_dim[l], kmult));
        h->Npad.push_back(pad_to(out_dim[l], amx::PAD_NT));
        if (l + 1 < m->n_layers)
            h->max_hidden_pad = std::max(h->max_hidden_pad, h->Npad[l]);
        long fl = (long)m->in_dim[l] * out_dim[l];
        if (fl > best_flops) {
            best_flops       = fl;
            h->largest_layer = l;
        }
    }
    for (int l = 0; l < m->n_layers; ++l) {
        const int    K = h->in[l], N = h->out[l], Kp = h->Kpad[l], Np = h->Npad[l];
        const float* W = Wl[l];
        void*        d = nullptr;
        if (prec == AMX_PREC_F16MX) {
            for (size_t i = 0; i < (size_t)N * K; ++i)
                if (std::fabs(W[i]) >= 65520.f) {  // not a number a trained layer holds; f16 cannot
                    amx::set_error("amx_ffnn_create: layer %d holds a weight outside the f16 range (%g): use AMX_PREC_BF16X3", l, (double)W[i]);
                    h->d_W.push_back(nullptr);
                    amx_ffnn_destroy(h);
                    return AMX_ERR_INVALID;
                }
            std::vector<unsigned char> pk;
            amx::mx::pack_weights_host(W, N, K, K, Np, Kp / 32, pk);
            if (hipMalloc(&d, pk.size()) != hipSuccess || hipMemcpy(d, pk.data(), pk.size(), hipMemcpyHostToDevice) != hipSuccess) {
                amx::set_error("amx_ffnn_create: device allocation of layer %d failed", l);
                h->d_W.push_back(d);
                amx_ffnn_destroy(h);
                return AMX_ERR_DEVICE;
            }
        }
        else if (prec == AMX_PREC_BF16X3) {
            // rows [W_hi | W_lo], each plane Kp columns wide (zero padded); the rows that feed the layer are [X_hi | X_lo] with the
            // lo plane at column xlo = Kpad (layer 0) or Npad of the layer below
            std::vector<amx::bf16_t> pk((size_t)Np * 2 * Kp, 0);
            for (int n = 0; n < N; ++n)
                for (int k = 0; k < K; ++k) {
                    const float       w  = W[(size_t)n * K + k];
                    const amx::bf16_t hi = amx::f2bf_host(w);
                    unsigned          hu = (unsigned)hi << 16;
                    float             hf;
                    memcpy(&hf, &hu, 4);
                    pk[(size_t)n * 2 * Kp + k]      = hi;
                    pk[(size_t)n * 2 * Kp + Kp + k] = amx::f2bf_host(w - hf);
                }
            if (hipMalloc(&d, pk.size() * 2) != hipSuccess || hipMemcpy(d, pk.data(), pk.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
                amx::set_error("amx_ffnn_create: device allocation of layer %d failed", l);
                h->d_W.push_back(d);
                amx_ffnn_destroy(h);
                return AMX_ERR_DEVICE;
            }
        }
        else if (prec == AMX_PREC_BF16) {
            std::vector<amx::bf16_t> pk((size_t)Np * Kp, 0);
            for (int n = 0; n < N; ++n)
                for (int k = 0; k < K; ++k)
                    pk[(size_t)n * Kp + k] = amx::f2bf_host(W[(size_t)n * K + k]);
            if (hipMalloc(&d, pk.size() * 2) != hipSuccess || hipMemcpy(d, pk.data(), pk.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
                amx::set_error("amx_ffnn_create: device allocation of layer %d failed", l);
                h->d_W.push_back(d);
                amx_ffnn_destroy(h);
                return AMX_ERR_DEVICE;
            }
        }
        else {
            std::vector<float> pk((size_t)Np * Kp, 0.f);
            for (int n = 0; n < N; ++n)
                memcpy(&pk[(size_t)n * Kp], W + (size_t)n * K, (size_t)K * 4);
            if (hipMalloc(&d, pk.size() * 4) != hipSuccess || hipMemcpy(d, pk.data(), pk.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                amx::set_error("amx_ffnn_create: device allocation of layer %d failed", l);
                h->d_W.push_back(d);
                amx_ffnn_destroy(h);
                return AMX_ERR_DEVICE;
            }
        }
        h->d_W.push_back(d);
        std::vector<float> b((size_t)Np, 0.f);
        for (int n = 0; n < N; ++n) {
            float v = bl[l] ? bl[l][n] : 0.f;
            // removeLogPriorFromBias (Nn/LinearAndActivationLayer.hh:137-160): bias -= scale * prior
            if (l == m->n_layers - 1 && prior && m->prior_scale != 0.f) {
                float prod = m->prior_scale * prior[n];
                v          = v - prod;
            }
            b[n] = v;
        }
        if (l == m->n_layers - 1) {
            h->h_Wout.assign(W, W + (size_t)N * K);
            h->h_bout.assign(b.begin(), b.begin() + N);
        }
        float* db = nullptr;
        if (hipMalloc((void**)&db, b.size() * 4) != hipSuccess || hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            amx::set_error("amx_ffnn_create: device allocation of bias %d failed", l);
            h->d_bias.push_back(db);
            amx_ffnn_destroy(h);
            return AMX_ERR_DEVICE;
        }
        h->d_bias.push_back(db);
    }
    *out = h;
    return AMX_OK;
}

void amx_ffnn_destroy(amx_ffnn* h) {
    if (!h)
        return;
    hipSetDevice(h->ctx->device);
    for (void* p : h->d_W)
        hipFree(p);
    for (float* p : h->d_bias)
        hipFree(p);
    hipFree(h->d_in);
    hipFree(h->d_act[0]);
    hipFree(h->d_act[1]);
    hipFree(h->d_Wout);
    hipFree(h->d_bout);
    hipFree(h->d_rowstat);
    hipFree(h->d_part_min);
    hipFree(h->d_part_idx);
    hipFree(h->d_host_f);
    hipFree(h->d_host_s);
    if (h->h_overflow)
        hipHostFree(h->h_overflow);
    hipFree(h->d_ks_ws);
    for (auto& kv : h->graphs)
        if (kv.second)  // nullptr marks a signature seen once
            hipGraphExecDestroy(kv.second);
    delete h;
}

int amx_ffnn_input_dim(const amx_ffnn* h) {
    return h ? h->in[0] : 0;
}
int amx_ffnn_output_dim(const amx_ffnn* h) {
    return h ? h->out.back() : 0;
}

extern "C" int amx_stats_accumulate_dev(amx_ctx*, const float*, int, int, uint32_t*, unsigned long long*, double*);

// internal (not in amx.h): combine per-tile arg-min partials [n_tiles x part_ld]; shared with the GMM scorer's fused statistics
extern "C" int amx_internal_best_state_reduce(amx_ctx* ctx, const float* part_min, const unsigned* part_idx, int n_tiles, int part_ld, int T,
                                              uint32_t* best_state_dev, unsigned long long* counts_dev, double* score_sum_dev) {
    hipLaunchKernelGGL(amx::best_state_reduce_kernel, dim3((T + 63) / 64), dim3(256), 0, ctx->stream, part_min, part_idx, n_tiles, part_ld, T,
                       best_state_dev, counts_dev, score_sum_dev);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

static int ffnn_score_launches(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* scores_dev, bool stats,
                               uint32_t* best_state_dev, unsigned long long* counts_dev, double* score_sum_dev);

static int ffnn_score_impl(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* scores_dev, bool stats,
                           uint32_t* best_state_dev, unsigned long long* counts_dev, double* score_sum_dev) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_ffnn_score_dev: NULL handle");
    AMX_REQUIRE(T >= 0, AMX_ERR_INVALID, "amx_ffnn_score_dev: negative frame count");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_dev && scores_dev, AMX_ERR_INVALID, "amx_ffnn_score_dev: NULL buffer");
    AMX_REQUIRE(feats_stride >= h->in[0], AMX_ERR_INVALID, "amx_ffnn_score_dev: feature stride %d < input dimension %d", feats_stride, h->in[0]);
    AMX_REQUIRE(!h->overflowed(), AMX_ERR_STATE,
                "amx_ffnn_score_dev: a feature or hidden activation left the f16 range (|v| >= 65520) in an earlier pass of this AMX_PREC_F16MX "
                "handle; its scores were not valid -- create the scorer with AMX_PREC_BF16X3");
    AMX_HIP(hipSetDevice(h->ctx->device));
    // Small batches (the decoder's ring buffer: 256 ... 1024 frames, the same device buffers every time): replay the pass as a
    // HIP graph.  Not while profiling (the per-launch events are not part of the graph).
    const bool graphable = h->use_graphs && !h->ctx->profiling && T <= 4096 && h->mfma_bf16();
    if (!graphable)
        return ffnn_score_launches(h, feats_dev, feats_stride, T, scores_dev, stats, best_state_dev, counts_dev, score_sum_dev);
    const amx_ffnn::GraphKey key{feats_dev, scores_dev, best_state_dev, counts_dev, score_sum_dev, h->ctx->stream, feats_stride, T, stats ? 1 : 0};
    auto                     it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        // first call with this signature: run it plainly once (sizes the workspace, sets kernel attributes), capture the second time.
        // The cap is checked HERE: a caller that slides its pointers through a large buffer never repeats a signature, and its
        // "seen once" entries would otherwise grow the map for the life of the handle.
        if (h->graphs.size() >= 64) {
            for (auto& kv : h->graphs)
                if (kv.second)
                    hipGraphExecDestroy(kv.second);
            h->graphs.clear();
            h->use_graphs = 0;
            return ffnn_score_launches(h, feats_dev, feats_stride, T, scores_dev, stats, best_state_dev, counts_dev, score_sum_dev);
        }
        static const hipGraphExec_t kSeenOnce = nullptr;
        h->graphs[key] = kSeenOnce;
        return ffnn_score_launches(h, feats_dev, feats_stride, T, scores_dev, stats, best_state_dev, counts_dev, score_sum_dev);
    }
    if (it->second == nullptr) {
        if (h->graphs.size() > 64) {  // a caller that keeps changing buffers: stop caching
            h->use_graphs = 0;
            return ffnn_score_launches(h, feats_dev, feats_stride, T, scores_dev, stats, best_state_dev, counts_dev, score_sum_dev);
        }
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(h->ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            h->use_graphs = 0;
            return ffnn_score_launches(h, feats_dev, feats_stride, T, scores_dev, stats, best_state_dev, counts_dev, score_sum_dev);
        }
        const int  r   = ffnn_score_launches(h, feats_dev, feats_stride, T, scores_dev, stats, best_state_dev, counts_dev, score_sum_dev);
        const bool ok  = hipStreamEndCapture(h->ctx->stream, &g) == hipSuccess && r == AMX_OK && g != nullptr;
        hipGraphExec_t ex = nullptr;
        if (!ok || hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (g)
                hipGraphDestroy(g);
            h->use_graphs = 0;  // capture is not available on this stream: plain launches from now on
            return ffnn_score_launches(h, feats_dev, feats_stride, T, scores_dev, stats, best_state_dev, counts_dev, score_sum_dev);
        }
        hipGraphDestroy(g);
        it->second = ex;
    }
    AMX_HIP(hipGraphLaunch(it->second, h->ctx->stream));
    return AMX_OK;
}

// hidden_out != nullptr: run the hidden layers only and export the last hidden activation as f32 [T x hidden_dim]
static int ffnn_launches(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* scores_dev, bool stats,
                         uint32_t* best_state_dev, unsigned long long* counts_dev, double* score_sum_dev, float* hidden_out);

static int ffnn_score_launches(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* scores_dev, bool stats,
                               uint32_t* best_state_dev, unsigned long long* counts_dev, double* score_sum_dev) {
    return ffnn_launches(h, feats_dev, feats_stride, T, scores_dev, stats, best_state_dev, counts_dev, score_sum_dev, nullptr);
}

static int ffnn_launches(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* scores_dev, bool stats,
                         uint32_t* best_state_dev, unsigned long long* counts_dev, double* score_sum_dev, float* hidden_out) {
    const int chunk = h->chunk;  // frames per pass (workspace: 2 x chunk x max_hidden x 2 B)
    const int L     = h->n_layers;
    for (int t0 = 0; t0 < T; t0 += chunk) {
        const int Tc   = std::min(chunk, T - t0);
        const int Tpad = pad_to(Tc, amx::PAD_NT);
        int       r    = ensure_workspace(h, Tpad);
        if (r != AMX_OK)
            return r;
        const float* x = feats_dev + (size_t)t0 * feats_stride;
        {
            amx::ScopedKernelTimer timer(h->ctx, "ffnn_pack");
            const int blocks = (int)std::min<long long>(4096, ((long long)Tpad * h->Kpad[0] + 255) / 256);
            if (h->is_mx()) {
                const int b2 = (int)std::min<long long>(8192, ((long long)Tpad * (h->Kpad[0] / 32) * 2 + 255) / 256);
                hipLaunchKernelGGL(amx::mx::pack_input_mx, dim3(b2), dim3(256), 0, h->ctx->stream, x, feats_stride, Tc, h->in[0], (char*)h->d_in,
                                   h->Kpad[0] / 32, Tpad, h->d_overflow);
            }
            else if (h->precision == AMX_PREC_BF16X3)
                hipLaunchKernelGGL(amx::pack_input_bf16x3, dim3(blocks), dim3(256), 0, h->ctx->stream, x, feats_stride, Tc, h->in[0],
                                   (amx::bf16_t*)h->d_in, h->Kpad[0], Tpad);
            else if (h->precision == AMX_PREC_BF16)
                hipLaunchKernelGGL(amx::pack_input_bf16, dim3(blocks), dim3(256), 0, h->ctx->stream, x, feats_stride, Tc, h->in[0],
                                   (amx::bf16_t*)h->d_in, h->Kpad[0], Tpad);
            else
                hipLaunchKernelGGL(amx::pack_input_f32, dim3(blocks), dim3(256), 0, h->ctx->stream, x, feats_stride, Tc, h->in[0],
                                   (float*)h->d_in, h->Kpad[0], Tpad);
            AMX_HIP(hipGetLastError());
        }
        const void* cur = h->d_in;
        const bool  x3  = h->precision == AMX_PREC_BF16X3;
        const int   planes = x3 ? 2 : 1;  // split bf16: rows are [hi plane | lo plane]
        int         ldx    = h->is_mx() ? h->Kpad[0] / 32 : planes * h->Kpad[0];  // f16mx: K-tiles per 256-row block of the operand
        const bool  fused = stats && h->mfma_bf16();
        h->cur_part_min = nullptr;
        h->cur_part_idx = nullptr;
        if (fused) {
            const size_t need = (size_t)(h->Npad[L - 1] / 128) * Tpad;  // >= n-tiles of any configuration
            if (need > h->part_cap) {
                for (auto& kv : h->graphs)  // captured passes hold the old addresses
                    if (kv.second)
                        hipGraphExecDestroy(kv.second);
                h->graphs.clear();
                hipFree(h->d_part_min);
                hipFree(h->d_part_idx);
                h->d_part_min = nullptr;
                h->d_part_idx = nullptr;
                h->part_cap   = 0;
                AMX_HIP(hipMalloc((void**)&h->d_part_min, need * 4));
                AMX_HIP(hipMalloc((void**)&h->d_part_idx, need * 4));
                h->part_cap = need;
            }
            h->cur_part_min = h->d_part_min;
            h->cur_part_idx = h->d_part_idx;
        }
        for (int l = 0; l < L; ++l) {
            if (h->is_mx() && hidden_out && l >= L - 2) {
                // the on-demand scorer's f32 hidden activation: the last hidden layer leaves through the score epilogue as
                // -(W x + b) in f32 (exact), act(-v) restores the activation; a network without hidden layers exports its input
                const int H = h->in[L - 1];
                float*    dst = hidden_out + (size_t)t0 * H;
                if (L == 1) {
                    const int blocks = (int)std::min<long long>(8192, ((long long)Tc * H + 255) / 256);
                    hipLaunchKernelGGL(amx::export_hidden_kernel<0>, dim3(blocks), dim3(256), 0, h->ctx->stream, (const void*)x, feats_stride, 0, Tc, H, dst);
                }
                else if (l == L - 2) {
                    r = launch_layer<true>(h, l, cur, ldx, dst, H, Tc, Tpad);
                    if (r != AMX_OK)
                        return r;
                    const long long n = (long long)Tc * H;
                    const int blocks = (int)std::min<long long>(8192, (n + 255) / 256);
                    switch (h->act[l]) {
                        case AMX_ACT_RELU: hipLaunchKernelGGL(amx::mx::neg_act_kernel<AMX_ACT_RELU>, dim3(blocks), dim3(256), 0, h->ctx->stream, dst, n); break;
                        case AMX_ACT_SIGMOID: hipLaunchKernelGGL(amx::mx::neg_act_kernel<AMX_ACT_SIGMOID>, dim3(blocks), dim3(256), 0, h->ctx->stream, dst, n); break;
                        case AMX_ACT_TANH: hipLaunchKernelGGL(amx::mx::neg_act_kernel<AMX_ACT_TANH>, dim3(blocks), dim3(256), 0, h->ctx->stream, dst, n); break;
                        default: hipLaunchKernelGGL(amx::mx::neg_act_kernel<AMX_ACT_NONE>, dim3(blocks), dim3(256), 0, h->ctx->stream, dst, n); break;
                    }
                }
                AMX_HIP(hipGetLastError());
                if (L == 1 || l == L - 2)
                    break;
                continue;
            }
            if (l == L - 1 && hidden_out) {
                const int H = h->in[l];  // = out[l - 1], or the input dimension of a network without hidden layers
                float*    dst = hidden_out + (size_t)t0 * H;
                const int blocks = (int)std::min<long long>(8192, ((long long)Tc * H + 255) / 256);
                if (x3)
                    hipLaunchKernelGGL(amx::export_hidden_kernel<2>, dim3(blocks), dim3(256), 0, h->ctx->stream, cur, ldx, ldx / 2, Tc, H, dst);
                else if (h->precision == AMX_PREC_BF16)
                    hipLaunchKernelGGL(amx::export_hidden_kernel<1>, dim3(blocks), dim3(256), 0, h->ctx->stream, cur, ldx, 0, Tc, H, dst);
                else
                    hipLaunchKernelGGL(amx::export_hidden_kernel<0>, dim3(blocks), dim3(256), 0, h->ctx->stream, cur, ldx, 0, Tc, H, dst);
                AMX_HIP(hipGetLastError());
            }
            else if (l == L - 1) {
                float* sc = scores_dev + (size_t)t0 * h->out[l];
                r         = launch_layer<true>(h, l, cur, ldx, sc, h->out[l], Tc, Tpad);
                if (r == AMX_OK && fused) {
                    amx::ScopedKernelTimer timer(h->ctx, "stats");
                    hipLaunchKernelGGL(amx::best_state_reduce_kernel, dim3((Tc + 63) / 64), dim3(256), 0, h->ctx->stream,
                                       h->d_part_min, h->d_part_idx, h->cur_ntn, Tpad, Tc, best_state_dev ? best_state_dev + t0 : nullptr,
                                       counts_dev, score_sum_dev);
                    AMX_HIP(hipGetLastError());
                }
                else if (r == AMX_OK && stats)  // fp32 parity path: separate arg-min pass over the scores
                    r = amx_stats_accumulate_dev(h->ctx, sc, Tc, h->out[l], best_state_dev ? best_state_dev + t0 : nullptr, counts_dev,
                                                 score_sum_dev);
            }
            else {
                void* dst = h->d_act[l & 1];  // split bf16: the epilogue applies the activation and writes both planes
                r         = launch_layer<false>(h, l, cur, ldx, dst, planes * h->Npad[l], Tc, Tpad);
                cur       = dst;
                ldx       = h->is_mx() ? h->Npad[l] / 32 : planes * h->Npad[l];
            }
            if (r != AMX_OK)
                return r;
        }
    }
    h->cur_part_min = nullptr;
    h->cur_part_idx = nullptr;
    return AMX_OK;
}

static const char* const kOverflowText =
        "%s: a feature or hidden activation left the f16 range (|v| >= 65520, inf or NaN) in a pass of this AMX_PREC_F16MX handle; the scores of "
        "that pass are not valid -- create the scorer with AMX_PREC_BF16X3";

int amx_ffnn_precision(const amx_ffnn* h, double* mx_block_ratio) {
    if (!h)
        return AMX_ERR_INVALID;
    if (mx_block_ratio)
        *mx_block_ratio = h->mx_block_ratio;
    return h->precision;
}

int amx_ffnn_wait_dev(amx_ffnn* h) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_ffnn_wait_dev: NULL handle");
    AMX_REQUIRE(h->ctx, AMX_ERR_STATE, "amx_ffnn_wait_dev: host-only handle");
    AMX_HIP(hipSetDevice(h->ctx->device));
    AMX_HIP(hipStreamSynchronize(h->ctx->stream));
    AMX_REQUIRE(!h->overflowed(), AMX_ERR_STATE, kOverflowText, "amx_ffnn_wait_dev");
    return AMX_OK;
}

int amx_ffnn_hidden_dim(const amx_ffnn* h) {
    return h ? h->in.back() : 0;
}

int amx_ffnn_forward_hidden_dev(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* act_dev) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_ffnn_forward_hidden_dev: NULL handle");
    AMX_REQUIRE(T >= 0, AMX_ERR_INVALID, "amx_ffnn_forward_hidden_dev: negative frame count");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_dev && act_dev, AMX_ERR_INVALID, "amx_ffnn_forward_hidden_dev: NULL buffer");
    AMX_REQUIRE(feats_stride >= h->in[0], AMX_ERR_INVALID, "amx_ffnn_forward_hidden_dev: feature stride %d < input dimension %d", feats_stride,
                h->in[0]);
    AMX_REQUIRE(!h->overflowed(), AMX_ERR_STATE, kOverflowText, "amx_ffnn_forward_hidden_dev");
    AMX_HIP(hipSetDevice(h->ctx->device));
    return ffnn_launches(h, feats_dev, feats_stride, T, nullptr, false, nullptr, nullptr, nullptr, act_dev);
}

int amx_ffnn_score_on_demand_dev(amx_ffnn* h, const float* act_dev, int n_pairs, const uint32_t* frame_dev, const uint32_t* emission_dev,
                                 float* scores_dev) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_ffnn_score_on_demand_dev: NULL handle");
    AMX_REQUIRE(n_pairs >= 0, AMX_ERR_INVALID, "amx_ffnn_score_on_demand_dev: negative pair count");
    if (n_pairs == 0)
        return AMX_OK;
    AMX_REQUIRE(act_dev && frame_dev && emission_dev && scores_dev, AMX_ERR_INVALID, "amx_ffnn_score_on_demand_dev: NULL buffer");
    AMX_REQUIRE(!h->overflowed(), AMX_ERR_STATE, kOverflowText, "amx_ffnn_score_on_demand_dev");   // its hidden activations come from this handle
    AMX_HIP(hipSetDevice(h->ctx->device));
    if (!h->d_Wout) {  // OnDemandFeatureScorer::init pops the output layer and keeps its parameters apart: upload them on first use
        AMX_HIP(hipMalloc((void**)&h->d_Wout, h->h_Wout.size() * 4));
        AMX_HIP(hipMalloc((void**)&h->d_bout, h->h_bout.size() * 4));
        AMX_HIP(hipMemcpy(h->d_Wout, h->h_Wout.data(), h->h_Wout.size() * 4, hipMemcpyHostToDevice));
        AMX_HIP(hipMemcpy(h->d_bout, h->h_bout.data(), h->h_bout.size() * 4, hipMemcpyHostToDevice));
    }
    amx::ScopedKernelTimer timer(h->ctx, "ffnn_on_demand");
    hipLaunchKernelGGL(amx::on_demand_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, h->ctx->stream, act_dev, h->in.back(), h->d_Wout, h->d_bout,
                       frame_dev, emission_dev, n_pairs, scores_dev);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_precomputed_score_dev(amx_ctx* ctx, const float* feats_dev, int feats_stride, int T, int n_classes, const int* class_to_output_dev,
                              const float* log_prior_dev, float prior_scale, float* scores_dev) {
    AMX_REQUIRE(ctx, AMX_ERR_INVALID, "amx_precomputed_score_dev: NULL context");
    AMX_REQUIRE(T >= 0 && n_classes > 0, AMX_ERR_INVALID, "amx_precomputed_score_dev: bad shape");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_dev && log_prior_dev && scores_dev, AMX_ERR_INVALID, "amx_precomputed_score_dev: NULL buffer");
    AMX_HIP(hipSetDevice(ctx->device));
    amx::ScopedKernelTimer timer(ctx, "precomputed_score");
    const int              blocks = (int)std::min<long long>(16384, ((long long)T * n_classes + 255) / 256);
    hipLaunchKernelGGL(amx::precomputed_score_kernel, dim3(blocks), dim3(256), 0, ctx->stream, feats_dev, feats_stride, T, n_classes,
                       class_to_output_dev, log_prior_dev, prior_scale, scores_dev);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_ffnn_score_dev(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* scores_dev) {
    return ffnn_score_impl(h, feats_dev, feats_stride, T, scores_dev, false, nullptr, nullptr, nullptr);
}

int amx_ffnn_forward_dev(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* out_dev, int top) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_ffnn_forward_dev: NULL handle");
    AMX_REQUIRE(top == AMX_NN_TOP_LINEAR || top == AMX_NN_TOP_SOFTMAX, AMX_ERR_INVALID, "amx_ffnn_forward_dev: unknown top layer mode %d", top);
    AMX_REQUIRE(!h->class_mapped, AMX_ERR_INVALID,
                "amx_ffnn_forward_dev: the handle carries a class-label mapping; the forward node emits the network's own outputs");
    const int r = ffnn_score_impl(h, feats_dev, feats_stride, T, out_dev, false, nullptr, nullptr, nullptr);
    if (r != AMX_OK || T == 0)
        return r;
    const int       n = h->out.back();
    const long long total = (long long)T * n;
    hipStream_t     st = h->ctx->stream;
    amx::ScopedKernelTimer timer(h->ctx, "ffnn_top");
    const int blocks = (int)std::min<long long>(16384, (total + 255) / 256);
    if (top == AMX_NN_TOP_LINEAR) {
        hipLaunchKernelGGL(amx::top_finish_kernel, dim3(blocks), dim3(256), 0, st, out_dev, total, n, (const float*)nullptr);
        AMX_HIP(hipGetLastError());
        return AMX_OK;
    }
    if (T > h->rowstat_cap) {
        hipFree(h->d_rowstat);
        h->d_rowstat   = nullptr;
        h->rowstat_cap = 0;
        AMX_HIP(hipMalloc((void**)&h->d_rowstat, (size_t)2 * T * 4));
        h->rowstat_cap = T;
    }
    hipLaunchKernelGGL(amx::top_negexp_kernel, dim3(T), dim3(256), 0, st, out_dev, n, h->d_rowstat);
    hipLaunchKernelGGL(amx::top_rowsum_kernel, dim3((T + 63) / 64), dim3(64), 0, st, out_dev, T, n, h->d_rowstat + h->rowstat_cap);
    hipLaunchKernelGGL(amx::top_finish_kernel, dim3(blocks), dim3(256), 0, st, out_dev, total, n, (const float*)(h->d_rowstat + h->rowstat_cap));
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

int amx_ffnn_score_stats_dev(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* scores_dev, uint32_t* best_state_dev,
                             unsigned long long* state_counts_dev, double* score_sum_dev) {
    AMX_REQUIRE(state_counts_dev && score_sum_dev, AMX_ERR_INVALID, "amx_ffnn_score_stats_dev: NULL accumulator");
    return ffnn_score_impl(h, feats_dev, feats_stride, T, scores_dev, true, best_state_dev, state_counts_dev, score_sum_dev);
}

int amx_ffnn_score(amx_ffnn* h, const float* feats_host, int T, float* scores_host) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_ffnn_score: NULL handle");
    AMX_REQUIRE(T >= 0, AMX_ERR_INVALID, "amx_ffnn_score: negative frame count");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_host && scores_host, AMX_ERR_INVALID, "amx_ffnn_score: NULL buffer");
    AMX_HIP(hipSetDevice(h->ctx->device));
    // staging buffers live in the handle and only grow (no hipMalloc / hipFree per call; unchanged device addresses also let
    // ffnn_score_impl replay its HIP graph)
    hipStream_t  st = h->ctx->stream;
    const size_t nf = (size_t)T * h->in[0], ns = (size_t)T * h->out.back();
    auto grow = [h](float** p, size_t* cap, size_t need) {
        if (need <= *cap)
            return true;
        for (auto& kv : h->graphs)  // captured passes may hold the old staging addresses: their keys would never be hit again
            if (kv.second)
                hipGraphExecDestroy(kv.second);
        h->graphs.clear();
        hipFree(*p);
        *p   = nullptr;
        *cap = 0;
        if (hipMalloc((void**)p, need * 4) != hipSuccess)
            return false;
        *cap = need;
        return true;
    };
    if (!grow(&h->d_host_f, &h->host_f_cap, nf) || !grow(&h->d_host_s, &h->host_s_cap, ns)) {
        (void)hipGetLastError();
        amx::set_error("amx_ffnn_score: out of device memory");
        return AMX_ERR_DEVICE;
    }
    if (hipMemcpyAsync(h->d_host_f, feats_host, nf * 4, hipMemcpyHostToDevice, st) != hipSuccess) {
        amx::set_error("amx_ffnn_score: H2D copy failed");
        return AMX_ERR_DEVICE;
    }
    int r = amx_ffnn_score_dev(h, h->d_host_f, h->in[0], T, h->d_host_s);
    if (r != AMX_OK)
        return r;
    if (hipMemcpyAsync(scores_host, h->d_host_s, ns * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        amx::set_error("amx_ffnn_score: D2H copy / kernel execution failed: %s", hipGetErrorString(hipGetLastError()));
        return AMX_ERR_DEVICE;
    }
    AMX_REQUIRE(!h->overflowed(), AMX_ERR_STATE,
                "amx_ffnn_score: a feature or hidden activation left the f16 range (|v| >= 65520); the scores are not valid -- create the "
                "scorer with AMX_PREC_BF16X3");
    return AMX_OK;
}

}  // extern "C"

#ifdef AMX_LAB
extern "C" int amx_lab_mx_tile_stamps(unsigned long long* out /* [4 tiles][3 phases][memtime, memrealtime] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(amx::mx::mx_tile_stamps), sizeof(amx::mx::mx_tile_stamps)) == hipSuccess ? AMX_OK : AMX_ERR_DEVICE;
}
extern "C" int amx_lab_mx_stamps(unsigned long long* out /* [8 * 48 * 4] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(amx::mx::mx_stamps), sizeof(amx::mx::mx_stamps)) == hipSuccess ? AMX_OK : AMX_ERR_DEVICE;
}
#endif
