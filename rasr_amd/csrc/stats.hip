// stats.hip -- per-epoch statistics (amx_stats_*): best state per frame, per-state frame counts
// and the sum of best scores.  These are the accumulators a data-parallel job all-reduces once
// per epoch; in RASR they live in per-partition accumulator files that
// `acoustic-model-trainer --action=combine-mixture-set-estimators` sums offline
// (Tools/AcousticModelTrainer/AcousticModelTrainer.cc:317-325).
#include "common.hpp"

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
