// common.hpp -- internals shared by the librasr_amd.so translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/amx.h"

namespace amx {

void set_error(const char* fmt, ...);

#define AMX_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            amx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return AMX_ERR_DEVICE;                                                             \
        }                                                                                      \
    } while (0)

#define AMX_REQUIRE(cond, status, ...)   \
    do {                                 \
        if (!(cond)) {                   \
            amx::set_error(__VA_ARGS__); \
            return (status);             \
        }                                \
    } while (0)

// Per-kernel event timing (amx_profile_*): pairs of events recorded around a launch on the
// context's current stream; resolved lazily in amx_profile_get.
struct ProfileSlot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double                                          total_ms = 0;
    long                                            n        = 0;
};

}  // namespace amx

struct amx_ctx {
    int                                      device     = 0;
    hipStream_t                              own_stream = nullptr;
    hipStream_t                              stream     = nullptr;
    bool                                     profiling  = false;
    std::map<std::string, amx::ProfileSlot>  prof;
    int                                      n_cu = 0;
    int                                      contract = AMX_CONTRACT_OFF;   // amx_set_contract: which build of the reference the f32 arithmetic follows
    // scratch for amx_stats_accumulate_dev
    void*  scratch       = nullptr;
    size_t scratch_bytes = 0;

    int ensure_scratch(size_t bytes);
};

namespace amx {

// RAII-less helper: records start/stop events around a launch when profiling is on.
struct ScopedKernelTimer {
    amx_ctx*    ctx;
    const char* name;
    hipEvent_t  a = nullptr, b = nullptr;
    ScopedKernelTimer(amx_ctx* c, const char* n)
            : ctx(c), name(n) {
        if (ctx->profiling) {
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a, ctx->stream);
        }
    }
    ~ScopedKernelTimer() {
        if (a) {
            hipEventRecord(b, ctx->stream);
            ctx->prof[name].events.emplace_back(a, b);
        }
    }
};

// The `tuning` string of amx_mfcc_cfg / amx_gmm_model / amx_ffnn_model: "key=value,key=value".  A handle parses it once at
// creation against the list of keys it knows (an unknown key fails the creation: a typo must not silently select the default
// kernel) and keeps the values; nothing in the library reads kernel-selecting switches from the environment.  Lab builds
// (-DAMX_LAB, tools/ab_*.sh) append the AMX_TUNING environment variable to every handle's string, unknown keys ignored there.
struct Tuning {
    std::map<std::string, std::string> kv;
    // returns false (error text set) on a malformed string or a key that is not in `allowed` (NULL-terminated list)
    bool parse(const char* s, const char* const* allowed, const char* who);
    bool has(const char* key) const { return kv.count(key) != 0; }
    // Typed reads, checked at creation: false (error text set) unless the value is a whole decimal number within [lo, hi] /
    // one of `words` (NULL-terminated).  A value the key does not take fails the creation like an unknown key does.
    bool get_int(const char* key, int dflt, long lo, long hi, int* out, const char* who) const;
    bool get_word(const char* key, const char* dflt, const char* const* words, std::string* out, const char* who) const;
};

// keys of amx_gmm_model.tuning (gmm.hip and gmm_simd.hip parse the same string)
static const char* const gmm_tuning_keys[] = {"screen", "fused", "screen_all", "screen_kernel", "graph", "tied_prune", "chunk", "fused_waves", "fr",
                                              "simd_mfma", "contract", "dist_list", "near_fused", "fused_pack", nullptr};

// the ONE source of a fused multiply-add outside the GMM scorers (gmm_device.hpp has sq_acc<FMA>): a * b + c as the reference's
// default build computes it at a contracted site (FMA) or with two roundings (the library is compiled with -ffp-contract=off)
template<bool FMA>
__host__ __device__ __forceinline__ float mad(float a, float b, float c) {
    return FMA ? __builtin_fmaf(a, b, c) : a * b + c;
}
template<bool FMA>
__host__ __device__ __forceinline__ double mad(double a, double b, double c) {
    return FMA ? __builtin_fma(a, b, c) : a * b + c;
}

// streaming stores of the score / best-density matrices (written once, read by a later kernel or the host): non-temporal, so that
// gigabytes of results do not push the operand panels out of L2.  -DAMX_PLAIN_STORES (tools/energy_table.sh: a measurement build)
// turns them into plain stores for the energy / time comparison of profiles/r06/energy.json.
template<class V>
__device__ __forceinline__ void nt_store(V v, V* p) {
#ifdef AMX_PLAIN_STORES
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}

inline int ceil_div(long a, long b) {
    return (int)((a + b - 1) / b);
}

}  // namespace amx
