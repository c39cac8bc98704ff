// cgo_comm.hip — cross-rank exchange of the per-launch scalar block.
//
// The reference has no communication layer at all.  Sharding
// the state vector adds exactly one exchange per reduction point: an
// all-gather of NS doubles (80 B per rank) followed by a rank-ordered local
// sum, so every rank holds bitwise identical scalars and runs the identical
// line-search state machine without any broadcast of decisions.
//
//   RcclComm      ncclAllGather on the ctx stream over xGMI (one process per GPU).
//                 librccl is dlopen()ed lazily so single-GPU use never needs it.
//   CallbackComm  the host supplies the all-gather (e.g. MPI.jl, or gloo in tests).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "cgo_hip_backend.hpp"

namespace cgo {

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi &api() {
    static RcclApi a;
    if (a.handle) return a;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        a.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (a.handle) break;
    }
    if (!a.handle) return a;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
    a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.AllGather && a.CommDestroy && a.GetErrorString;
    return a;
}

struct RcclComm : Comm {
    ncclComm_t comm = nullptr;
    int device = 0;
    ~RcclComm() override {
        if (comm) {
            (void)hipSetDevice(device);
            api().CommDestroy(comm);
        }
    }
    int allgather_host(const double *, double *, int) override { return -1; }
    int allgather_device(const double *send, double *recv, int count, void *stream) override {
        ncclResult_t r = api().AllGather(send, recv, (size_t)count, ncclFloat64, comm, (hipStream_t)stream);
        if (r != ncclSuccess) {
            set_error(std::string("ncclAllGather: ") + api().GetErrorString(r));
            return 1;
        }
        return 0;
    }
};

struct CallbackComm : Comm {
    cgo_allgather_fn fn = nullptr;
    void *user = nullptr;
    int allgather_host(const double *send, double *recv, int count) override {
        return fn(user, send, recv, count);
    }
};

}  // namespace

int rccl_unique_id(void *out128) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    RcclApi &a = api();
    if (!a.ok) { set_error("librccl could not be loaded"); return CGO_ECOMM; }
    ncclUniqueId id;
    ncclResult_t r = a.GetUniqueId(&id);
    if (r != ncclSuccess) { set_error(std::string("ncclGetUniqueId: ") + a.GetErrorString(r)); return CGO_ECOMM; }
    std::memcpy(out128, &id, 128);
    return CGO_OK;
}

Comm *make_rccl_comm(HipCtx *ctx, int rank, int world, const void *unique_id128) {
    RcclApi &a = api();
    if (!a.ok) { set_error("librccl could not be loaded"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, 128);
    RcclComm *c = new RcclComm();
    c->rank = rank; c->world = world; c->device = ctx->device;
    ncclResult_t r = a.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        set_error(std::string("ncclCommInitRank: ") + a.GetErrorString(r));
        c->comm = nullptr;
        delete c;
        return nullptr;
    }
    return c;
}

Comm *make_callback_comm(int rank, int world, cgo_allgather_fn fn, void *user) {
    CallbackComm *c = new CallbackComm();
    c->rank = rank; c->world = world; c->fn = fn; c->user = user;
    return c;
}

}  // namespace cgo
