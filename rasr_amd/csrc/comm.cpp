// comm.cpp -- the ONE cross-rank exchange of the path: the per-epoch sum of the accumulators (amx_comm_* of include/amx.h).
//
// RASR has no communication backend: data-parallel trainers are independent processes over `partition` / `select-partition`
// (Bliss/CorpusDescription.cc:174-190) that write accumulator files, and `combine-mixture-set-estimators`
// (Tools/AcousticModelTrainer/AcousticModelTrainer.cc:317-325 -> Mm/AbstractMixtureSetEstimator.cc:173-250) adds them up offline.
// Here the ranks of one node keep their accumulators in HBM as ONE flat f64 buffer and add them with one RCCL all-reduce over
// xGMI.  RCCL is bound at run time (dlopen): the library loads, and every other entry point works, on a box without it.
#include "common.hpp"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace {

// the part of RCCL's C API this file needs (rccl.h: ncclUniqueId is 128 opaque bytes; enums as in nccl.h 2.x)
struct RcclUniqueId {
    char internal[AMX_COMM_ID_BYTES];
};
typedef void* RcclComm;
enum { kRcclSuccess = 0, kRcclFloat64 = 8, kRcclSum = 0 };

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId*)                                                       = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int)                                  = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t)           = nullptr;
    int (*CommDestroy)(RcclComm)                                                            = nullptr;
    const char* (*GetErrorString)(int)                                                      = nullptr;
    int (*GetVersion)(int*)                                                                 = nullptr;
    std::string error;
};

RcclApi& rccl() {
    static RcclApi   api;
    static std::once_flag once;
    std::call_once(once, [] {
        // AMX_RCCL_LIB names the library explicitly; otherwise the loader's search path (an already loaded librccl -- e.g. the one
        // a PyTorch process carries -- is found by its soname), then the ROCm default location
        const char* names[] = {getenv("AMX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            if (!n || !*n)
                continue;
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle)
                break;
            const char* e = dlerror();
            api.error     = e ? e : "dlopen failed";
        }
        if (!api.handle) {
            if (api.error.empty())
                api.error = "librccl.so not found";
            return;
        }
        api.GetUniqueId    = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
        api.CommInitRank   = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
        api.AllReduce      = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
        api.CommDestroy    = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
        api.GetVersion     = (decltype(api.GetVersion))dlsym(api.handle, "ncclGetVersion");
        if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) {
            api.error  = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
            api.handle = nullptr;
        }
    });
    return api;
}

const char* rccl_error(int rc) {
    RcclApi& r = rccl();
    return r.GetErrorString ? r.GetErrorString(rc) : "unknown RCCL error";
}

}  // namespace

struct amx_comm {
    amx_ctx* ctx   = nullptr;
    RcclComm comm  = nullptr;
    int      rank  = 0;
    int      world = 1;
};

extern "C" {

int amx_comm_available(void) {
    return rccl().handle != nullptr;
}

int amx_comm_unique_id(unsigned char id[AMX_COMM_ID_BYTES]) {
    AMX_REQUIRE(id, AMX_ERR_INVALID, "amx_comm_unique_id: id is NULL");
    RcclApi& r = rccl();
    AMX_REQUIRE(r.handle, AMX_ERR_STATE, "amx_comm_unique_id: RCCL is not available (%s)", r.error.c_str());
    RcclUniqueId u;
    const int    rc = r.GetUniqueId(&u);
    AMX_REQUIRE(rc == kRcclSuccess, AMX_ERR_DEVICE, "amx_comm_unique_id: ncclGetUniqueId failed: %s", rccl_error(rc));
    memcpy(id, u.internal, AMX_COMM_ID_BYTES);
    return AMX_OK;
}

int amx_comm_init(amx_ctx* ctx, int rank, int world, const unsigned char id[AMX_COMM_ID_BYTES], amx_comm** out) {
    AMX_REQUIRE(ctx && id && out, AMX_ERR_INVALID, "amx_comm_init: NULL argument");
    *out = nullptr;
    AMX_REQUIRE(world >= 1 && rank >= 0 && rank < world, AMX_ERR_INVALID, "amx_comm_init: rank %d of %d", rank, world);
    RcclApi& r = rccl();
    AMX_REQUIRE(r.handle, AMX_ERR_STATE, "amx_comm_init: RCCL is not available (%s)", r.error.c_str());
    AMX_HIP(hipSetDevice(ctx->device));  // one rank per device: the communicator binds to the context's GPU
    RcclUniqueId u;
    memcpy(u.internal, id, AMX_COMM_ID_BYTES);
    RcclComm  c  = nullptr;
    const int rc = r.CommInitRank(&c, world, u, rank);
    AMX_REQUIRE(rc == kRcclSuccess && c, AMX_ERR_DEVICE, "amx_comm_init: ncclCommInitRank(rank %d of %d) failed: %s", rank, world,
                rccl_error(rc));
    amx_comm* h = new amx_comm;
    h->ctx      = ctx;
    h->comm     = c;
    h->rank     = rank;
    h->world    = world;
    *out        = h;
    return AMX_OK;
}

int amx_comm_rank(const amx_comm* c) {
    return c ? c->rank : -1;
}

int amx_comm_world(const amx_comm* c) {
    return c ? c->world : 0;
}

int amx_comm_all_reduce_f64_dev(amx_comm* c, double* buf_dev, size_t n) {
    AMX_REQUIRE(c, AMX_ERR_INVALID, "amx_comm_all_reduce_f64_dev: NULL communicator");
    if (n == 0)
        return AMX_OK;
    AMX_REQUIRE(buf_dev, AMX_ERR_INVALID, "amx_comm_all_reduce_f64_dev: NULL buffer");
    AMX_HIP(hipSetDevice(c->ctx->device));
    amx::ScopedKernelTimer timer(c->ctx, "all_reduce");
    const int              rc = rccl().AllReduce(buf_dev, buf_dev, n, kRcclFloat64, kRcclSum, c->comm, c->ctx->stream);
    AMX_REQUIRE(rc == kRcclSuccess, AMX_ERR_DEVICE, "amx_comm_all_reduce_f64_dev: ncclAllReduce(%zu doubles) failed: %s", n, rccl_error(rc));
    return AMX_OK;
}

void amx_comm_destroy(amx_comm* c) {
    if (!c)
        return;
    hipSetDevice(c->ctx->device);
    hipStreamSynchronize(c->ctx->stream);
    if (c->comm)
        rccl().CommDestroy(c->comm);
    delete c;
}

}  // extern "C"
