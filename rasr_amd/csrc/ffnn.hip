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
#include <cmath>
#include <cstring>

namespace amx {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short                              bf16_t;

typedef float  f32x2 __attribute__((ext_vector_type(2)));
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
constexpr int BN = 128, BT = 128, BK = 64;
constexpr int TILE_BYTES = BN * BK * 2;  // 16 KB per operand tile

// byte offset of 16-byte chunk `c` (0..7) of row `r` inside a [128][64] bf16 tile
__device__ __forceinline__ int swz(int r, int c) {
    return r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
}

template<int ACT, bool LAST>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ X,
                                                          const float* __restrict__ bias, void* __restrict__ out, int Kpad, int ldx,
                                                          int ldo, int n_valid, int t_valid, int n_tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [2 buffers][W tile | X tile]
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn   = wave >> 1, wt = wave & 1;

    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8); give each
    // XCD a contiguous run of tiles that share the same frame block, so its L2 keeps that X panel.
    const int nwg = gridDim.x;
    int       id  = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int tile_t = id / n_tiles_n;
    const int tile_n = id - tile_t * n_tiles_n;
    const int n0 = tile_n * BN, t0 = tile_t * BT;

    // ---- staging: each wave moves 8 rows x 128 B per instruction; 4 instructions per operand
    // LDS linear position of this lane's 16 B within the tile for instruction i:
    //   pos = (i*4 + wave) * 1024 + lane*16  -> row = pos / 128, physical chunk = (pos % 128) / 16
    // it must hold logical chunk = phys ^ ((row>>1)&7) of that row.
    const bf16_t* gW[4];
    const bf16_t* gX[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pos  = (i * 4 + wave) * 1024 + lane * 16;
        const int row  = pos >> 7;
        const int phys = (pos & 127) >> 4;
        const int chunk = phys ^ ((row >> 1) & 7);
        gW[i] = W + (size_t)(n0 + row) * Kpad + chunk * 8;
        gX[i] = X + (size_t)(t0 + row) * ldx + chunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = lds + buf * (2 * TILE_BYTES);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const void*)(gW[i] + (size_t)kt * BK),
                                             (__attribute__((address_space(3))) void*)(base + (i * 4 + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const void*)(gX[i] + (size_t)kt * BK),
                                             (__attribute__((address_space(3))) void*)(base + TILE_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;

    const int KT = Kpad / BK;
    stage(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) {
            stage(buf ^ 1, kt + 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile kt landed, tile kt+1 (8 loads) in flight
        }
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* wbase = lds + buf * (2 * TILE_BYTES);
        const char* xbase = wbase + TILE_BYTES;
        const int   frow  = lane & 31;
        const int   fk    = lane >> 5;  // which 8-element half of the 16-wide k slab
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wn * 64 + i * 32 + frow;
                a[i]        = *(const bf16x8*)(wbase + swz(r, ks * 2 + fk));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wt * 64 + j * 32 + frow;
                b[j]        = *(const bf16x8*)(xbase + swz(r, ks * 2 + fk));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        // all of this wave's LDS reads have returned before it signals that buf may be restaged
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: lane holds, per 32x32 tile, col t = lane&31 and rows n = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int t = t0 + wt * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * (lane >> 5);
                float     v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = acc[i][j][g * 4 + e] + bias[n + e];  // addToAllColumns
                if (LAST) {
                    if (t < t_valid) {
                        float* o = (float*)out + (size_t)t * ldo + n;
                        if (n + 3 < n_valid && ((ldo & 3) == 0))
                            *(float4*)o = make_float4(-v[0], -v[1], -v[2], -v[3]);
                        else
                            for (int e = 0; e < 4; ++e)
                                if (n + e < n_valid)
                                    o[e] = -v[e];
                    }
                }
                else {
                    uint2 pk;
                    pk.x = pack_bf16(activate<ACT>(v[0]), activate<ACT>(v[1]));
                    pk.y = pack_bf16(activate<ACT>(v[2]), activate<ACT>(v[3]));
                    *(uint2*)((bf16_t*)out + (size_t)t * ldo + n) = pk;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 GEMM (parity mode): exact f32 MFMA, register-staged LDS tiles with padded rows.
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
    int    max_hidden_pad = 0;
    int    largest_layer  = 0;
    size_t elt() const { return precision == AMX_PREC_BF16 ? 2 : 4; }
};

namespace {

int pad_to(int v, int m) {
    return (v + m - 1) / m * m;
}

int ensure_workspace(amx_ffnn* h, int Tpad) {
    if (Tpad <= h->cap_T)
        return AMX_OK;
    hipFree(h->d_in);
    hipFree(h->d_act[0]);
    hipFree(h->d_act[1]);
    h->d_in = h->d_act[0] = h->d_act[1] = nullptr;
    h->cap_T                            = 0;
    AMX_HIP(hipMalloc(&h->d_in, (size_t)Tpad * h->Kpad[0] * h->elt()));
    if (h->max_hidden_pad > 0) {
        AMX_HIP(hipMalloc(&h->d_act[0], (size_t)Tpad * h->max_hidden_pad * h->elt()));
        AMX_HIP(hipMalloc(&h->d_act[1], (size_t)Tpad * h->max_hidden_pad * h->elt()));
    }
    h->cap_T = Tpad;
    return AMX_OK;
}

template<bool LAST>
int launch_layer(amx_ffnn* h, int l, const void* x, int ldx, void* out, int ldo, int T, int Tpad) {
    const int   ntn = h->Npad[l] / amx::BN, ntt = Tpad / amx::BT;
    dim3        grid(ntn * ntt), block(256);
    hipStream_t st = h->ctx->stream;
    const int   nv = h->out[l];
    amx::ScopedKernelTimer t_all(h->ctx, "ffnn_gemm");
    hipEvent_t             e0 = nullptr, e1 = nullptr;
    const bool             time_max = h->ctx->profiling && l == h->largest_layer;
    if (time_max) {
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, st);
    }
    if (h->precision == AMX_PREC_BF16) {
        const size_t lds = 2 * 2 * amx::TILE_BYTES;
#define AMX_L(ACT)                                                                                                     \
    {                                                                                                                  \
        auto k = amx::gemm_bf16_kernel<ACT, LAST>;                                                                      \
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
        hipLaunchKernelGGL(k, grid, block, lds, st, (const amx::bf16_t*)h->d_W[l], (const amx::bf16_t*)x, h->d_bias[l], \
                           out, h->Kpad[l], ldx, ldo, nv, T, ntn);                                                      \
    }
        switch (LAST ? AMX_ACT_NONE : h->act[l]) {
            case AMX_ACT_RELU: AMX_L(AMX_ACT_RELU) break;
            case AMX_ACT_SIGMOID: AMX_L(AMX_ACT_SIGMOID) break;
            case AMX_ACT_TANH: AMX_L(AMX_ACT_TANH) break;
            default: AMX_L(AMX_ACT_NONE) break;
        }
#undef AMX_L
    }
    else {
#define AMX_L(ACT)                                                                                                   \
    hipLaunchKernelGGL((amx::gemm_f32_kernel<ACT, LAST>), grid, block, 0, st, (const float*)h->d_W[l], (const float*)x, \
                       h->d_bias[l], (float*)out, h->Kpad[l], ldx, ldo, nv, T, ntn);
        switch (LAST ? AMX_ACT_NONE : h->act[l]) {
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
    AMX_REQUIRE(m->precision == AMX_PREC_FP32 || m->precision == AMX_PREC_BF16, AMX_ERR_INVALID, "amx_ffnn_create: unknown precision");
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

    amx_ffnn* h  = new amx_ffnn;
    h->ctx       = ctx;
    h->n_layers  = m->n_layers;
    h->precision = m->precision;
    hipSetDevice(ctx->device);
    const int kmult = (m->precision == AMX_PREC_BF16) ? amx::BK : amx::FK;
    long      best_flops = -1;
    for (int l = 0; l < m->n_layers; ++l) {
        h->in.push_back(m->in_dim[l]);
        h->out.push_back(m->out_dim[l]);
        h->act.push_back(m->activation[l]);
        // hidden activations are stored with a row stride of Npad(l-1) >= Kpad(l)
        h->Kpad.push_back(pad_to(m->in_dim[l], kmult));
        h->Npad.push_back(pad_to(m->out_dim[l], amx::BN));
        if (l + 1 < m->n_layers)
            h->max_hidden_pad = std::max(h->max_hidden_pad, h->Npad[l]);
        long fl = (long)m->in_dim[l] * m->out_dim[l];
        if (fl > best_flops) {
            best_flops       = fl;
            h->largest_layer = l;
        }
    }
    for (int l = 0; l < m->n_layers; ++l) {
        const int    K = h->in[l], N = h->out[l], Kp = h->Kpad[l], Np = h->Npad[l];
        const float* W = m->W[l];
        void*        d = nullptr;
        if (m->precision == AMX_PREC_BF16) {
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
            float v = m->bias[l] ? m->bias[l][n] : 0.f;
            // removeLogPriorFromBias (Nn/LinearAndActivationLayer.hh:137-160): bias -= scale * prior
            if (l == m->n_layers - 1 && m->log_prior && m->prior_scale != 0.f) {
                float prod = m->prior_scale * m->log_prior[n];
                v          = v - prod;
            }
            b[n] = v;
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
    delete h;
}

int amx_ffnn_input_dim(const amx_ffnn* h) {
    return h ? h->in[0] : 0;
}
int amx_ffnn_output_dim(const amx_ffnn* h) {
    return h ? h->out.back() : 0;
}

int amx_ffnn_score_dev(amx_ffnn* h, const float* feats_dev, int feats_stride, int T, float* scores_dev) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_ffnn_score_dev: NULL handle");
    AMX_REQUIRE(T >= 0, AMX_ERR_INVALID, "amx_ffnn_score_dev: negative frame count");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_dev && scores_dev, AMX_ERR_INVALID, "amx_ffnn_score_dev: NULL buffer");
    AMX_REQUIRE(feats_stride >= h->in[0], AMX_ERR_INVALID, "amx_ffnn_score_dev: feature stride %d < input dimension %d", feats_stride, h->in[0]);
    AMX_HIP(hipSetDevice(h->ctx->device));
    const int chunk = 32768;  // frames per pass (workspace: 2 x 32768 x max_hidden x 2 B)
    const int L     = h->n_layers;
    for (int t0 = 0; t0 < T; t0 += chunk) {
        const int Tc   = std::min(chunk, T - t0);
        const int Tpad = pad_to(Tc, amx::BT);
        int       r    = ensure_workspace(h, Tpad);
        if (r != AMX_OK)
            return r;
        const float* x = feats_dev + (size_t)t0 * feats_stride;
        {
            amx::ScopedKernelTimer timer(h->ctx, "ffnn_pack");
            const int blocks = (int)std::min<long long>(4096, ((long long)Tpad * h->Kpad[0] + 255) / 256);
            if (h->precision == AMX_PREC_BF16)
                hipLaunchKernelGGL(amx::pack_input_bf16, dim3(blocks), dim3(256), 0, h->ctx->stream, x, feats_stride, Tc, h->in[0],
                                   (amx::bf16_t*)h->d_in, h->Kpad[0], Tpad);
            else
                hipLaunchKernelGGL(amx::pack_input_f32, dim3(blocks), dim3(256), 0, h->ctx->stream, x, feats_stride, Tc, h->in[0],
                                   (float*)h->d_in, h->Kpad[0], Tpad);
            AMX_HIP(hipGetLastError());
        }
        const void* cur = h->d_in;
        int         ldx = h->Kpad[0];
        for (int l = 0; l < L; ++l) {
            if (l == L - 1)
                r = launch_layer<true>(h, l, cur, ldx, scores_dev + (size_t)t0 * h->out[l], h->out[l], Tc, Tpad);
            else {
                void* dst = h->d_act[l & 1];
                r         = launch_layer<false>(h, l, cur, ldx, dst, h->Npad[l], Tc, Tpad);
                cur       = dst;
                ldx       = h->Npad[l];
            }
            if (r != AMX_OK)
                return r;
        }
    }
    return AMX_OK;
}

int amx_ffnn_score(amx_ffnn* h, const float* feats_host, int T, float* scores_host) {
    AMX_REQUIRE(h, AMX_ERR_INVALID, "amx_ffnn_score: NULL handle");
    AMX_REQUIRE(T >= 0, AMX_ERR_INVALID, "amx_ffnn_score: negative frame count");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(feats_host && scores_host, AMX_ERR_INVALID, "amx_ffnn_score: NULL buffer");
    AMX_HIP(hipSetDevice(h->ctx->device));
    float *      d_f = nullptr, *d_s = nullptr;
    hipStream_t  st = h->ctx->stream;
    const size_t nf = (size_t)T * h->in[0], ns = (size_t)T * h->out.back();
    auto         done = [&](int code) {
        hipFree(d_f);
        hipFree(d_s);
        return code;
    };
    if (hipMalloc((void**)&d_f, nf * 4) != hipSuccess || hipMalloc((void**)&d_s, ns * 4) != hipSuccess) {
        amx::set_error("amx_ffnn_score: out of device memory");
        return done(AMX_ERR_DEVICE);
    }
    if (hipMemcpyAsync(d_f, feats_host, nf * 4, hipMemcpyHostToDevice, st) != hipSuccess) {
        amx::set_error("amx_ffnn_score: H2D copy failed");
        return done(AMX_ERR_DEVICE);
    }
    int r = amx_ffnn_score_dev(h, d_f, h->in[0], T, d_s);
    if (r != AMX_OK)
        return done(r);
    if (hipMemcpyAsync(scores_host, d_s, ns * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        amx::set_error("amx_ffnn_score: D2H copy / kernel execution failed: %s", hipGetErrorString(hipGetLastError()));
        return done(AMX_ERR_DEVICE);
    }
    return done(AMX_OK);
}

}  // extern "C"
