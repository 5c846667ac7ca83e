// stats.hip -- per-epoch statistics (amx_stats_*): best state per frame, per-state frame counts
// and the sum of best scores.  These are the accumulators a data-parallel job all-reduces once
// per epoch; in RASR they live in per-partition accumulator files that
// `acoustic-model-trainer --action=combine-mixture-set-estimators` sums offline
// (Tools/AcousticModelTrainer/AcousticModelTrainer.cc:317-325).
#include "common.hpp"

#include <algorithm>
#include <cfloat>

namespace amx {

// one wavefront per frame: strided scan over the emissions, first minimum wins
__global__ __launch_bounds__(256) void argmin_accumulate_kernel(const float* __restrict__ scores, int T, int M,
                                                               uint32_t* __restrict__ best_state,
                                                               unsigned long long* __restrict__ counts, double* __restrict__ score_sum) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double    local_sum = 0.0;
    uint32_t           run_idx = 0xffffffffu;
    unsigned long long run_len = 0;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < T; t += (long long)gridDim.x * 4) {
        const float* row  = scores + (size_t)t * M;
        float        best = FLT_MAX;
        uint32_t     idx  = 0xffffffffu;
        for (int e = lane; e < M; e += 64) {
            float v = row[e];
            if (v < best) {  // ascending e within a lane: keeps the first minimum
                best = v;
                idx  = (uint32_t)e;
            }
        }
        // butterfly reduction; ties resolved towards the smaller emission index
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            float    ob = __shfl_xor(best, off, 64);
            uint32_t oi = (uint32_t)__shfl_xor((int)idx, off, 64);
            if (ob < best || (ob == best && oi < idx)) {
                best = ob;
                idx  = oi;
            }
        }
        if (lane == 0) {
            if (best_state)
                best_state[t] = idx;
            if (idx != 0xffffffffu) {
                // run-length aggregation: consecutive frames of this wave that pick the same state share one atomic
                if (idx == run_idx)
                    ++run_len;
                else {
                    if (run_len)
                        atomicAdd(&counts[run_idx], run_len);
                    run_idx = idx;
                    run_len = 1;
                }
                local_sum += (double)best;
            }
        }
    }
    if (lane == 0 && run_len)
        atomicAdd(&counts[run_idx], run_len);
    if (lane == 0 && local_sum != 0.0)
        atomicAdd(score_sum, local_sum);
}

}  // namespace amx

namespace amx {
// scores[rows[i]][cols[i]] -> out[i]: the decoder's active (frame, emission) pairs of a device-resident score block
__global__ __launch_bounds__(256) void gather_scores_kernel(const float* __restrict__ scores, int ld, int n, const uint32_t* __restrict__ rows,
                                                           const uint32_t* __restrict__ cols, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] = scores[(size_t)rows[i] * ld + cols[i]];
}
}  // namespace amx

// ---- device-buffer plumbing for C / C++ clients that keep score blocks resident (rasr_amd/host/BatchFeatureScorer.hh): the
// reference's scorers hand the decoder ONE score at a time (Mm::FeatureScorer::ContextScorer::score, Mm/FeatureScorer.hh:31-46), so
// the [bufferSize x nEmissions] block stays in HBM and only the rows / pairs the decoder asks for cross PCIe.
extern "C" int amx_device_malloc(amx_ctx* ctx, size_t bytes, void** dev) {
    AMX_REQUIRE(ctx && dev, AMX_ERR_INVALID, "amx_device_malloc: NULL argument");
    *dev = nullptr;
    AMX_HIP(hipSetDevice(ctx->device));
    AMX_HIP(hipMalloc(dev, bytes ? bytes : 1));
    return AMX_OK;
}

extern "C" void amx_device_free(amx_ctx* ctx, void* dev) {
    if (!ctx || !dev)
        return;
    hipSetDevice(ctx->device);
    hipFree(dev);
}

extern "C" int amx_copy_to_device(amx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    AMX_REQUIRE(ctx && (bytes == 0 || (dst_dev && src_host)), AMX_ERR_INVALID, "amx_copy_to_device: NULL argument");
    AMX_HIP(hipSetDevice(ctx->device));
    if (!bytes)
        return AMX_OK;
    // The contract is "the source may be reused on return".  A pageable source is staged by the runtime before hipMemcpyAsync
    // returns; a pinned one (hipHostMalloc / hipHostRegister, e.g. a torch pinned tensor) is read by the DMA engine later, so the
    // copy is waited for.
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, src_host) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned)
        (void)hipGetLastError();  // "not a registered pointer" is the expected answer for pageable memory
    AMX_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (pinned)
        AMX_HIP(hipStreamSynchronize(ctx->stream));
    return AMX_OK;
}

extern "C" int amx_copy_to_host(amx_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    AMX_REQUIRE(ctx && (bytes == 0 || (dst_host && src_dev)), AMX_ERR_INVALID, "amx_copy_to_host: NULL argument");
    AMX_HIP(hipSetDevice(ctx->device));
    if (bytes)
        AMX_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AMX_HIP(hipStreamSynchronize(ctx->stream));
    return AMX_OK;
}

extern "C" int amx_gather_scores(amx_ctx* ctx, const float* scores_dev, int n_rows, int ld, int n, const uint32_t* rows_host,
                                 const uint32_t* cols_host, float* dst_host) {
    AMX_REQUIRE(ctx, AMX_ERR_INVALID, "amx_gather_scores: NULL context");
    AMX_REQUIRE(n >= 0 && ld > 0 && n_rows >= 0, AMX_ERR_INVALID, "amx_gather_scores: bad shape");
    if (n == 0)
        return AMX_OK;
    AMX_REQUIRE(scores_dev && rows_host && cols_host && dst_host, AMX_ERR_INVALID, "amx_gather_scores: NULL buffer");
    for (int i = 0; i < n; ++i)  // the decoder's indices are caller data: an index outside the resident block must not become a wild device read
        AMX_REQUIRE(rows_host[i] < (uint32_t)n_rows && cols_host[i] < (uint32_t)ld, AMX_ERR_INVALID,
                    "amx_gather_scores: pair %d = (row %u, column %u) outside the %d x %d block", i, rows_host[i], cols_host[i], n_rows, ld);
    AMX_HIP(hipSetDevice(ctx->device));
    int r = ctx->ensure_scratch((size_t)n * 12);
    if (r != AMX_OK)
        return r;
    uint32_t* d_rows = (uint32_t*)ctx->scratch;
    uint32_t* d_cols = d_rows + n;
    float*    d_out  = (float*)(d_cols + n);
    AMX_HIP(hipMemcpyAsync(d_rows, rows_host, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    AMX_HIP(hipMemcpyAsync(d_cols, cols_host, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(amx::gather_scores_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, scores_dev, ld, n, d_rows, d_cols, d_out);
    AMX_HIP(hipGetLastError());
    AMX_HIP(hipMemcpyAsync(dst_host, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    AMX_HIP(hipStreamSynchronize(ctx->stream));
    return AMX_OK;
}

extern "C" int amx_stats_accumulate_dev(amx_ctx* ctx, const float* scores_dev, int T, int n_emissions, uint32_t* best_state_dev,
                                        unsigned long long* state_counts_dev, double* score_sum_dev) {
    AMX_REQUIRE(ctx, AMX_ERR_INVALID, "amx_stats_accumulate_dev: NULL context");
    AMX_REQUIRE(T >= 0 && n_emissions > 0, AMX_ERR_INVALID, "amx_stats_accumulate_dev: bad shape");
    if (T == 0)
        return AMX_OK;
    AMX_REQUIRE(scores_dev && state_counts_dev && score_sum_dev, AMX_ERR_INVALID, "amx_stats_accumulate_dev: NULL buffer");
    AMX_HIP(hipSetDevice(ctx->device));
    const int              blocks = (int)std::min<long long>(((long long)T + 3) / 4, 8192);
    amx::ScopedKernelTimer timer(ctx, "stats");
    hipLaunchKernelGGL(amx::argmin_accumulate_kernel, dim3(blocks), dim3(256), 0, ctx->stream, scores_dev, T, n_emissions,
                       best_state_dev, state_counts_dev, score_sum_dev);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

// ---- u64 counters <-> f64 slots of the flat per-epoch reduce buffer (amx_comm_all_reduce_f64_dev sums doubles)
namespace amx {
__global__ __launch_bounds__(256) void counts_to_f64_kernel(const unsigned long long* __restrict__ c, double* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (double)c[i];
}
__global__ __launch_bounds__(256) void f64_to_counts_kernel(const double* __restrict__ in, unsigned long long* __restrict__ c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        c[i] = (unsigned long long)llrint(in[i]);
}
}  // namespace amx

extern "C" int amx_counts_to_f64_dev(amx_ctx* ctx, const unsigned long long* counts_dev, double* out_dev, size_t n) {
    AMX_REQUIRE(ctx, AMX_ERR_INVALID, "amx_counts_to_f64_dev: NULL context");
    if (n == 0)
        return AMX_OK;
    AMX_REQUIRE(counts_dev && out_dev, AMX_ERR_INVALID, "amx_counts_to_f64_dev: NULL buffer");
    AMX_HIP(hipSetDevice(ctx->device));
    const int blocks = (int)std::min<size_t>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(amx::counts_to_f64_kernel, dim3(blocks), dim3(256), 0, ctx->stream, counts_dev, out_dev, n);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

extern "C" int amx_f64_to_counts_dev(amx_ctx* ctx, const double* in_dev, unsigned long long* counts_dev, size_t n) {
    AMX_REQUIRE(ctx, AMX_ERR_INVALID, "amx_f64_to_counts_dev: NULL context");
    if (n == 0)
        return AMX_OK;
    AMX_REQUIRE(counts_dev && in_dev, AMX_ERR_INVALID, "amx_f64_to_counts_dev: NULL buffer");
    AMX_HIP(hipSetDevice(ctx->device));
    const int blocks = (int)std::min<size_t>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(amx::f64_to_counts_kernel, dim3(blocks), dim3(256), 0, ctx->stream, in_dev, counts_dev, n);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

// ---- the two device clocks, sampled in stream order (measurement: bench.py's epoch run reports the shader clock the chip sustains
// under the job's load -- real-time factor as Speech/CorpusProcessor.cc:49-58 defines it needs wall time only, but a throughput
// figure from 20 steps says nothing about the clock a 100 h epoch ends at)
namespace amx {
__global__ void device_clocks_kernel(unsigned long long* __restrict__ out) {
    if (threadIdx.x == 0) {
        out[0] = __builtin_amdgcn_s_memtime();      // shader clock ticks
        out[1] = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz counter
    }
}
// one workgroup per XCD (block b of a grid runs on XCD b % 8 -- observed, and not relied on: every block files its sample under the
// XCC id it reads from the hardware register; an XCD no block reached keeps its zeros).  The eight XCDs have a clock each.
__global__ void device_clocks_xcd_kernel(unsigned long long* __restrict__ out) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;  // HW_REG_XCC_ID
        out[2 * xcc]     = __builtin_amdgcn_s_memtime();
        out[2 * xcc + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
}  // namespace amx

extern "C" int amx_device_clocks_xcd_dev(amx_ctx* ctx, unsigned long long* out_dev) {
    AMX_REQUIRE(ctx && out_dev, AMX_ERR_INVALID, "amx_device_clocks_xcd_dev: NULL argument");
    AMX_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(amx::device_clocks_xcd_kernel, dim3(8), dim3(64), 0, ctx->stream, out_dev);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}

extern "C" int amx_device_clocks_dev(amx_ctx* ctx, unsigned long long* out_dev) {
    AMX_REQUIRE(ctx && out_dev, AMX_ERR_INVALID, "amx_device_clocks_dev: NULL argument");
    AMX_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(amx::device_clocks_kernel, dim3(1), dim3(64), 0, ctx->stream, out_dev);
    AMX_HIP(hipGetLastError());
    return AMX_OK;
}
