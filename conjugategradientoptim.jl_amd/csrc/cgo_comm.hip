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
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <ctime>

#include "cgo_hip_backend.hpp"

namespace cgo {

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
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
    a.CommCount = (decltype(a.CommCount))dlsym(a.handle, "ncclCommCount");
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
    int kind() const override { return 2; }
    int ranks_seen() override {
        int c = 0;
        if (!comm || !api().CommCount || api().CommCount(comm, &c) != ncclSuccess) return 0;
        return c;
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

// Host shared-memory mailbox (single node).  For ≤ 512-byte blocks the exchange is pure latency;
// a POSIX shm segment that every rank maps and registers with HIP lets each GPU's finalize kernel
// store {block, seq} directly where all hosts can read it: no collective call, no extra kernel,
// ≈ one PCIe write of latency.  Double-buffered on the launch sequence number.
// Device mailboxes (round 3; SURVEY.md §8(e) "fast path"): beside its host slots every rank owns a small block of ITS OWN
// HBM — [2 buffers][world][SLOT] doubles, fine-grained so that a peer's stores are visible to a running kernel — exports it
// with hipIpcGetMemHandle through the segment's header, and opens every peer's.  A GPU then stores its per-launch block
// straight into its peers' memory (over xGMI between GPUs; IPC works between processes on one GPU as well, which is how it
// is tested here) and polls its own mailbox: the exchange needs no host, so controller-armed launches work across ranks.
constexpr int XW_MAX = 8;   // ranks a launch argument block carries mailbox pointers for (one node)
struct ShmHeader {          // one per rank, behind the host slots
    unsigned char handle[64];            // hipIpcMemHandle_t of the rank's device mailbox
    unsigned long long exported;         // 0: not yet, 1: handle valid, 2: this rank has no device mailbox
    unsigned long long opened;           // 0: not yet, 1: this rank opened every peer's mailbox, 2: it could not (or gave up waiting)
    unsigned long long agreed;           // 0: not yet, 1: this rank's final verdict is "usable", 2: "unusable"
    unsigned long long pad[5];
};
static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle does not fit the header");

struct ShmComm : Comm {
    std::string name;
    void *base = nullptr, *dev = nullptr;
    size_t bytes = 0, slots_bytes = 0;
    bool registered = false;
    int device = 0;
    static constexpr int SLOT = 72;  // doubles per (rank, buffer): 64 values + seq + padding
    double *dmail_own = nullptr;         // this rank's device mailbox
    double *dmail[XW_MAX] = {};          // every rank's mailbox as this device addresses it
    bool dmail_open[XW_MAX] = {};
    int connected = -1;                  // −1: connect_devices() not called yet, 0: unavailable, 1: usable
    ShmHeader *header(int r) { return (ShmHeader *)((char *)base + slots_bytes) + r; }
    ~ShmComm() override {
        (void)hipSetDevice(device);
        for (int r = 0; r < XW_MAX; ++r) if (dmail_open[r] && dmail[r]) (void)hipIpcCloseMemHandle(dmail[r]);
        if (dmail_own) (void)hipFree(dmail_own);
        if (registered) (void)hipHostUnregister(base);
        if (base) munmap(base, bytes);
    }
    double *dev_mailbox(int r) override { return (connected == 1 && r >= 0 && r < world) ? dmail[r] : nullptr; }
    // Collective.  Phase 1: wait until every rank has exported (or declared it cannot); phase 2: open the peers' handles and
    // say so; phase 3: usable iff EVERY rank opened everything — all ranks reach the same verdict, or the armed launches of
    // one would wait for blocks another never sends.
    int connect_devices() override {
        if (connected >= 0) return connected;
        connected = 0;
        if (world > XW_MAX) return 0;
        auto wait_all = [&](bool second) -> int {   // → 1 all ok, 0 somebody cannot, −1 timeout
            const double t0 = now_s();
            for (;;) {
                int ok = 0, bad = 0;
                for (int r = 0; r < world; ++r) {
                    const unsigned long long v = __atomic_load_n(second ? &header(r)->opened : &header(r)->exported, __ATOMIC_ACQUIRE);
                    ok += v == 1; bad += v == 2;
                }
                if (bad) return 0;
                if (ok == world) return 1;
                if (now_s() - t0 > 60.0) return -1;
                usleep(200);
            }
        };
        (void)hipSetDevice(device);
        int verdict = wait_all(false);
        bool mine = verdict == 1;
        if (mine) {
            for (int r = 0; r < world && mine; ++r) {
                if (r == rank) { dmail[r] = dmail_own; continue; }
                hipIpcMemHandle_t h;
                std::memcpy(&h, header(r)->handle, sizeof h);
                void *p = nullptr;
                if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); mine = false; break; }
                dmail[r] = (double *)p; dmail_open[r] = true;
            }
        }
        __atomic_store_n(&header(rank)->opened, mine ? 1ull : 2ull, __ATOMIC_RELEASE);
        if (verdict == 1) verdict = wait_all(true);
        // A rank whose wait ran out must not end up alone with "unusable" while a slow peer later sees all opened == 1 and arms
        // launches that wait for blocks this rank never sends: it overwrites its OWN word with "cannot" — which turns every
        // later reader's verdict to 0 — and everybody confirms through a third word that holds each rank's final verdict;
        // usable iff all of them say 1 (a reader that times out here as well treats the mailbox as unusable AND says so).
        const bool ok2 = verdict == 1 && mine;
        if (!ok2) __atomic_store_n(&header(rank)->opened, 2ull, __ATOMIC_RELEASE);
        __atomic_store_n(&header(rank)->agreed, ok2 ? 1ull : 2ull, __ATOMIC_RELEASE);
        int fin = ok2 ? 1 : 0;
        if (ok2) {
            const double t0 = now_s();
            for (;;) {
                int ok = 0, bad = 0;
                for (int r = 0; r < world; ++r) {
                    const unsigned long long v = __atomic_load_n(&header(r)->agreed, __ATOMIC_ACQUIRE);
                    ok += v == 1; bad += v == 2;
                }
                if (bad) { fin = 0; break; }
                if (ok == world) break;
                if (now_s() - t0 > 60.0) { fin = 0; __atomic_store_n(&header(rank)->agreed, 2ull, __ATOMIC_RELEASE); break; }
                usleep(200);
            }
        }
        connected = fin;
        return connected;
    }
    static double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    int kind() const override { return 1; }
    int ranks_seen() override {   // slots that have published at least one launch
        int c = 0;
        for (int r = 0; r < world; ++r) {
            bool any = false;
            for (int b = 0; b < 2; ++b)
                any = any || __atomic_load_n((unsigned long long *)(shm_slot_host(r, b) + 64), __ATOMIC_ACQUIRE) != 0;
            c += any ? 1 : 0;
        }
        return c;
    }
    int allgather_host(const double *, double *, int) override { return -1; }
    double *shm_slot_host(int r, int buf) override { return (double *)base + ((size_t)r * 2 + buf) * SLOT; }
    double *shm_slot_dev(int r, int buf) override { return (double *)dev + ((size_t)r * 2 + buf) * SLOT; }
};

}  // namespace

Comm *make_shm_comm(HipCtx *ctx, int rank, int world, const char *name, int create) {
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    ShmComm *c = new ShmComm();
    c->rank = rank; c->world = world; c->name = name; c->device = ctx->device;
    c->slots_bytes = ((size_t)world * 2 * ShmComm::SLOT * sizeof(double) + 127) & ~(size_t)127;
    c->bytes = (c->slots_bytes + (size_t)world * sizeof(ShmHeader) + 4095) & ~(size_t)4095;
    int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) { set_error(std::string("shm_open(") + name + ") failed"); delete c; return nullptr; }
    if (create && ftruncate(fd, (off_t)c->bytes) != 0) { set_error("ftruncate on the shm segment failed"); close(fd); delete c; return nullptr; }
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { set_error("mmap of the shm segment failed"); delete c; return nullptr; }
    c->base = p;
    if (create) std::memset(p, 0, c->bytes);
    if (hipHostRegister(p, c->bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
        set_error("hipHostRegister of the shm segment failed");
        delete c;
        return nullptr;
    }
    c->registered = true;
    if (hipHostGetDevicePointer(&c->dev, p, 0) != hipSuccess) { set_error("hipHostGetDevicePointer failed"); delete c; return nullptr; }
    // this rank's device mailbox, exported through the header (a failure here only means: host mailbox only)
    unsigned long long exported = 2;
    if (world <= XW_MAX && !getenv("CGO_NO_DEVICE_MAILBOX")) {
        const size_t mb = sizeof(double) * 2 * (size_t)world * ShmComm::SLOT;
        void *q = nullptr;
        // FINE-GRAINED or nothing: the peers' stores must become visible to a kernel of this rank that is already running and
        // polling; coarse-grained memory gives no such promise (the bounded poll would run out and the solve end in an error
        // where the host mailbox works) — so a failure here publishes "cannot" (exported = 2) and every rank stays on the host mailbox.
        hipError_t e = hipExtMallocWithFlags(&q, mb, hipDeviceMallocFinegrained);
        if (e != hipSuccess) { (void)hipGetLastError(); q = nullptr; }
        hipIpcMemHandle_t h;
        if (e == hipSuccess && hipMemset(q, 0, mb) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
            hipIpcGetMemHandle(&h, q) == hipSuccess) {
            c->dmail_own = (double *)q;
            std::memcpy(c->header(rank)->handle, &h, sizeof h);
            exported = 1;
        } else {
            (void)hipGetLastError();
            if (q) (void)hipFree(q);
        }
    }
    __atomic_store_n(&c->header(rank)->exported, exported, __ATOMIC_RELEASE);
    return c;
}

bool rccl_available() { return api().ok; }

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
