// api.cpp -- context, error reporting and kernel-timing part of the C ABI (include/amx.h).
#include "common.hpp"

#include <cstring>

namespace amx {

static thread_local char g_error[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

}  // namespace amx

int amx_ctx::ensure_scratch(size_t bytes) {
    if (bytes <= scratch_bytes)
        return AMX_OK;
    if (scratch)
        hipFree(scratch);
    scratch       = nullptr;
    scratch_bytes = 0;
    AMX_HIP(hipMalloc(&scratch, bytes));
    scratch_bytes = bytes;
    return AMX_OK;
}

extern "C" {

const char* amx_version(void) {
    return "rasr_amd 0.1 (gfx950)";
}

const char* amx_last_error(void) {
    return amx::g_error;
}

int amx_init(int device_ordinal, amx_ctx** out) {
    AMX_REQUIRE(out, AMX_ERR_INVALID, "amx_init: out is NULL");
    *out  = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        amx::set_error("amx_init: no HIP device visible (this library has no CPU fallback)");
        return AMX_ERR_DEVICE;
    }
    AMX_REQUIRE(device_ordinal >= 0 && device_ordinal < n, AMX_ERR_INVALID,
                "amx_init: device ordinal %d out of range [0,%d)", device_ordinal, n);
    AMX_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    AMX_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        amx::set_error("amx_init: device %d is %s; this library only carries gfx950 (MI355X) code objects",
                       device_ordinal, prop.gcnArchName);
        return AMX_ERR_DEVICE;
    }
    amx_ctx* c = new amx_ctx;
    c->device  = device_ordinal;
    c->n_cu    = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        amx::set_error("amx_init: hipStreamCreate failed");
        return AMX_ERR_DEVICE;
    }
    c->stream = c->own_stream;
    *out      = c;
    return AMX_OK;
}

void amx_destroy(amx_ctx* ctx) {
    if (!ctx)
        return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->prof)
        for (auto& ev : kv.second.events) {
            hipEventDestroy(ev.first);
            hipEventDestroy(ev.second);
        }
    if (ctx->scratch)
        hipFree(ctx->scratch);
    if (ctx->own_stream)
        hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int amx_set_stream(amx_ctx* ctx, void* hip_stream) {
    AMX_REQUIRE(ctx, AMX_ERR_INVALID, "amx_set_stream: ctx is NULL");
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return AMX_OK;
}

int amx_synchronize(amx_ctx* ctx) {
    AMX_REQUIRE(ctx, AMX_ERR_INVALID, "amx_synchronize: ctx is NULL");
    AMX_HIP(hipSetDevice(ctx->device));
    AMX_HIP(hipStreamSynchronize(ctx->stream));
    return AMX_OK;
}

int amx_profile_enable(amx_ctx* ctx, int enable) {
    AMX_REQUIRE(ctx, AMX_ERR_INVALID, "amx_profile_enable: ctx is NULL");
    ctx->profiling = enable != 0;
    return AMX_OK;
}

static int resolve(amx_ctx* ctx) {
    AMX_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->prof) {
        for (auto& ev : kv.second.events) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
                kv.second.total_ms += ms;
                kv.second.n += 1;
            }
            hipEventDestroy(ev.first);
            hipEventDestroy(ev.second);
        }
        kv.second.events.clear();
    }
    return AMX_OK;
}

int amx_profile_reset(amx_ctx* ctx) {
    AMX_REQUIRE(ctx, AMX_ERR_INVALID, "amx_profile_reset: ctx is NULL");
    int r = resolve(ctx);
    ctx->prof.clear();
    return r;
}

int amx_profile_get(amx_ctx* ctx, const char* kernel, double* avg_ms, long* n_launches) {
    AMX_REQUIRE(ctx && kernel, AMX_ERR_INVALID, "amx_profile_get: NULL argument");
    int r = resolve(ctx);
    if (r != AMX_OK)
        return r;
    auto it = ctx->prof.find(kernel);
    double ms = 0;
    long   n  = 0;
    if (it != ctx->prof.end() && it->second.n > 0) {
        ms = it->second.total_ms / (double)it->second.n;
        n  = it->second.n;
    }
    if (avg_ms)
        *avg_ms = ms;
    if (n_launches)
        *n_launches = n;
    return AMX_OK;
}

}  // extern "C"
