// cgo_hip_backend.hip — HBM-resident solver state and launch plumbing.
//
// HBM layout per rank (shard of n_local doubles each, 256-B aligned hipMalloc):
//   x      current iterate                (in place: x ← x + a*·u inside the next launch)
//   u      search direction               (in place: u ← −g + βu)
//   gA,gB  gradient double buffer         g / g⁺ rotate by pointer swap on accept
//   p0     objective parameter (D)        read-only
//   S,Y    L-BFGS ring, m × n_local each  (only for CGO_BETA_LBFGS)
// The reference's xp, info.x and its three per-iteration copies
// (src/engine/optim.jl:136,139,140) have no counterpart: xp lives in registers.
#include "cgo_hip_backend.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include "cgo_kernels.hip.hpp"
#include "cgo_kernels_lse.hip.hpp"
#include "cgo_kernels_cg.hip.hpp"
#include "cgo_kernels_chain.hip.hpp"
#include "cgo_kernels_resident.hip.hpp"

namespace cgo {

using namespace dev;

static void unpack_r(const double *s, int k, Scal *out, bool dir);
static double bytes_r(int obj_kind, int mode, int64_t n, bool has_param);

static inline double now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e9 + (double)ts.tv_nsec;
}
// How long a host wait on a publishing kernel or on a peer's mailbox slot may last before the solve is given up
// (a peer died, a collective hung): CGO_WAIT_TIMEOUT_S, default 120 s.
static double wait_timeout_ns() {
    static const double t = [] { const char *e = getenv("CGO_WAIT_TIMEOUT_S"); double v = e ? atof(e) : 0.0; return (v > 0.0 ? v : 120.0) * 1e9; }();
    return t;
}

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const char *get_error() { return g_err.c_str(); }

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            set_error(std::string("HIP error: ") + hipGetErrorString(e__) + " at " #expr); \
            return CGO_EHIP;                                                               \
        }                                                                                  \
    } while (0)

int DevBuf::alloc(size_t count) {
    release();
    if (count == 0) count = 1;
    HIPCHK(hipMalloc((void **)&p, count * sizeof(double)));
    n = count;
    return CGO_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
}

HipObjective::~HipObjective() {
    if (host_x) (void)hipHostFree(host_x);
    if (host_g) (void)hipHostFree(host_g);
}

// ---------------------------------------------------------------- ctx
int HipCtx::init(int dev_id) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device visible: this engine has no CPU fallback (needs an MI355X / gfx950)");
        return CGO_ENODEV;
    }
    if (dev_id < 0 || dev_id >= count) { set_error("device index out of range"); return CGO_EINVAL; }
    HIPCHK(hipSetDevice(dev_id));
    device = dev_id;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev_id));
    arch = prop.gcnArchName;
    num_cu = prop.multiProcessorCount;
    if (arch.find("gfx950") == std::string::npos) {
        set_error("device is " + arch + "; this library carries gfx950 code objects only");
        return CGO_ENODEV;
    }
    HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc((void **)&partials, sizeof(double) * MAX_GRID * NG));
    HIPCHK(hipMalloc((void **)&partials2, sizeof(double) * (MAX_GRID / 64) * NG));
    HIPCHK(hipMalloc((void **)&out_dev, sizeof(double) * NG));
    HIPCHK(hipMalloc((void **)&tickets, sizeof(unsigned int) * (TAIL_GROUP + 2)));
    HIPCHK(hipMemset(tickets, 0, sizeof(unsigned int) * (TAIL_GROUP + 2)));
    HIPCHK(hipMalloc((void **)&partials_f, sizeof(double) * MAX_GRID * NR7));
    HIPCHK(hipMalloc((void **)&partials2_f, sizeof(double) * TAIL_GROUP * NG));
    static_assert((TAIL_EMPTY >> 32) == (TAIL_EMPTY & 0xFFFFFFFFull), "filled with a 32-bit pattern");
    HIPCHK(hipMemsetD32((hipDeviceptr_t)partials_f, (int)(TAIL_EMPTY & 0xFFFFFFFFull), (size_t)MAX_GRID * NR7 * 2));
    HIPCHK(hipMemsetD32((hipDeviceptr_t)partials2_f, (int)(TAIL_EMPTY & 0xFFFFFFFFull), (size_t)TAIL_GROUP * NG * 2));
    HIPCHK(hipDeviceSynchronize());
    if (const char *e = getenv("CGO_FUSED_TAIL")) fused_tail = (e[0] != '0');
    if (const char *e = getenv("CGO_TAIL_STRICT")) tail_strict = (e[0] == '1');
    HIPCHK(hipHostMalloc((void **)&host_pinned, sizeof(double) * NG * 64, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&host_seq, 64, hipHostMallocDefault));
    *host_seq = 0;
    if (const char *e = getenv("CGO_HOST_PUBLISH")) host_publish = (e[0] != '0');
    if (const char *e = getenv("CGO_FORCE_GATHER")) force_gather = (e[0] == '1');
    HIPCHK(hipEventCreate(&ev0));
    HIPCHK(hipEventCreate(&ev1));
    return CGO_OK;
}

void HipCtx::xch_collect() {
    if (!xev_pending) return;
    float ms = 0;
    if (hipEventQuery(xev1) == hipSuccess && hipEventElapsedTime(&ms, xev0, xev1) == hipSuccess) {
        xch_dev_ms += ms; xch_dev_n++;
        xev_pending = false;
    }
}

int HipCtx::ensure_gather() {
    if (gather_dev) return CGO_OK;
    if (world() > 64) { set_error("world size > 64 unsupported"); return CGO_EINVAL; }
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMalloc((void **)&gather_dev, sizeof(double) * NG * world()));
    return CGO_OK;
}

HipCtx::~HipCtx() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    comm.reset();
    if (stream) (void)hipStreamSynchronize(stream);
    if (partials) (void)hipFree(partials);
    if (partials2) (void)hipFree(partials2);
    if (out_dev) (void)hipFree(out_dev);
    if (tickets) (void)hipFree(tickets);
    if (partials_f) (void)hipFree(partials_f);
    if (partials2_f) (void)hipFree(partials2_f);
    if (gather_dev) (void)hipFree(gather_dev);
    if (host_pinned) (void)hipHostFree(host_pinned);
    if (host_seq) (void)hipHostFree(host_seq);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (xev0) (void)hipEventDestroy(xev0);
    if (xev1) (void)hipEventDestroy(xev1);
    if (stream) (void)hipStreamDestroy(stream);
}

// ---------------------------------------------------------------- launch plumbing
// Streaming policy of a launch: BIG once the bytes it moves are far beyond the
// 256 MiB Infinity Cache (measured crossover between n = 1e7 and 1e8 for 7 streams).
// Launches that move more than this are pure HBM streams (contiguous chunks + non-temporal accesses);
// below it part of the working set is served by the 256 MiB Infinity Cache (grid-stride, default policy).
// Measured crossovers on MI355X (gpurun_out/big_threshold.log, k_cg family): read-only launches
// (3-point trial) gain from the streaming path above ≈ 0.45 GB per launch (0.6 GB: 109 vs 158 µs;
// 0.3 GB: 77 vs 66 µs), read-write launches above ≈ 1.4 GB (2 GB: 386 vs 410 µs; 1 GB: 205 vs 192 µs).
// `forced` > 0: the solver's policy (cgo_solver_policy::hbm_stream_bytes; the CGO_BIG_BYTES experiment override is folded
// into the policy when the solver is created — the kernel-level entry points, which have no solver, read it here).
static double big_bytes_for(double forced, bool read_only = false) {
    if (forced > 0.0) return forced;
    return read_only ? 4.5e8 : 1.4e9;
}
static double env_big_bytes() {
    static const double forced = [] { const char *e = getenv("CGO_BIG_BYTES"); double t = e ? atof(e) : 0.0; return t > 0.0 ? t : 0.0; }();
    return forced;
}
double HipBackend::big_bytes(bool read_only) const { return big_bytes_for(pol_.hbm_stream_bytes, read_only); }
static bool is_big(int obj_kind, int mode, int64_t n, bool hp, double forced) {
    const bool ro = (mode == M_UPG || mode == M_BETAONLY);
    return bytes_for(obj_kind, mode, n, hp) > big_bytes_for(forced, ro);
}

static int grid_capped(int64_t n, int cap) {
    static const int forced = [] { const char *e = getenv("CGO_GRID_SMALL"); int v = e ? atoi(e) : 0; return (v >= 1 && v <= MAX_GRID) ? v : 0; }();
    static const int per = [] { const char *e = getenv("CGO_GROUPS_PER_LANE"); int v = e ? atoi(e) : 2; return (v >= 1 && v <= 64) ? v : 2; }();
    if (forced) cap = forced;
    const int64_t n2 = n >> 1;
    int64_t blocks = (n2 + (int64_t)BLOCK * per - 1) / ((int64_t)BLOCK * per);
    if (blocks < 1) blocks = 1;
    if (blocks > cap) blocks = cap;
    return (int)blocks;
}

// grid-stride path, light kernels (≤ 4 streams, little arithmetic): 4 workgroups per CU
int grid_for(int64_t n) { return grid_capped(n, GRID_SMALL); }

// grid-stride path, k_cg family: one workgroup per CU while the working set is (mostly)
// Infinity-Cache resident — fewer partial rows and a shorter launch ramp beat extra waves for
// these heavier bodies (measured n = 1e6: 36.7 vs 42.3 µs/iteration, n = 1e7: 102.5 vs 110.4;
// scripts/small_n_sweep.sh) — but NOT for the light kernels (k_lbfgs_loop at n = 1e7: 102 vs 53 µs).
// The 5- and 7-point bodies carry 2–3× the FP64 work per byte: with one wave per SIMD (256 workgroups)
// loads, arithmetic and stores of a trip run back to back, two waves overlap them (n = 1.25e7, 7 points:
// 256 → 115.7 µs, 512 → 102.3, 1024 → 104.1, 2048 → 115.6; n = 2.5e7: 218 / 207 / 221 / 217;
// scripts/ab_grid.sh, gpurun_out/ab_grid.log).
// Since the launch carries its own reduction (finish_tail) one workgroup per CU is best up to n = 2e6 for them as well
// (7 points, events off, 256 vs 512 workgroups: n = 5e5 55.3k vs 51.0k it/s, 1e6 49.7k vs 47.2k, 2e6 37.9k vs 37.6k,
// 3e6 31.1k vs 32.0k, 1.25e7 10.2k vs 11.2k; scripts/r02_grid7.sh).
static int grid_cg(int64_t n, int npts = 1) {
    static const int cap57 = [] { const char *e = getenv("CGO_GRID_CG7"); int v = e ? atoi(e) : 0; return (v >= 1 && v <= MAX_GRID) ? v : 0; }();
    if (npts >= 5) return grid_capped(n, cap57 ? cap57 : (n <= 2000000 ? 256 : 512));
    // (extended Rosenbrock, 3 points, 256 / 512 / 1024 workgroups: n = 1e7 17.5k / 15.4k / 16.1k it/s, 2e7 7.8k / 8.4k / 7.7k; scripts/r02_grid3.sh)
    return grid_capped(n, n <= 16000000 ? 256 : 512);
}

// ALGORITHMIC bytes of one launch: 8·n·(distinct n-vectors read + written)
double bytes_for(int obj_kind, int mode, int64_t n, bool has_param) {
    const int p = (obj_kind == CGO_OBJ_QUAD_DIAG || has_param) ? 1 : 0;
    int v = 0;
    if (mode == M_INIT) v = 1 + p + 2;
    else if (mode == (M_TRIAL | M_BETA)) v = 3 + p + 1;
    else if (mode == M_TRIAL) v = 2 + p + 1;
    else if (mode == (M_ACCEPT | M_DIR | M_TRIAL | M_BETA)) v = 3 + p + 3;
    else if (mode == (M_ACCEPT | M_DIR)) v = 3 + 2;
    else if (mode == M_ACCEPT) v = 2 + 1;
    else if (mode == M_DIR) v = 2 + 1;
    else if (mode == M_RESET) v = 1 + 1;
    else if (mode == M_UPG) v = 2;
    else if (mode == M_BETAONLY) v = 3;
    return 8.0 * (double)n * (double)v;
}

template <class Obj, bool BIG>
static int launch_obj(int mode, const KParams &P, int grid, hipStream_t st) {
    switch (mode) {
    case M_INIT: k_fused<Obj, M_INIT, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case M_TRIAL | M_BETA: k_fused<Obj, M_TRIAL | M_BETA, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case M_TRIAL: k_fused<Obj, M_TRIAL, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case M_ACCEPT | M_DIR | M_TRIAL | M_BETA:
        k_fused<Obj, M_ACCEPT | M_DIR | M_TRIAL | M_BETA, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    default: return -1;
    }
    return 0;
}

template <bool BIG>
static int launch_any(int obj_kind, int mode, const KParams &P, int grid, hipStream_t st) {
    switch (mode) {  // objective-free modes
    case M_ACCEPT | M_DIR: k_fused<ObjQuadDiag, M_ACCEPT | M_DIR, BIG><<<grid, BLOCK, 0, st>>>(P); return 0;
    case M_ACCEPT: k_fused<ObjQuadDiag, M_ACCEPT, BIG><<<grid, BLOCK, 0, st>>>(P); return 0;
    case M_DIR: k_fused<ObjQuadDiag, M_DIR, BIG><<<grid, BLOCK, 0, st>>>(P); return 0;
    case M_RESET: k_fused<ObjQuadDiag, M_RESET, BIG><<<grid, BLOCK, 0, st>>>(P); return 0;
    case M_UPG: k_fused<ObjQuadDiag, M_UPG, BIG><<<grid, BLOCK, 0, st>>>(P); return 0;
    case M_BETAONLY: k_fused<ObjQuadDiag, M_BETAONLY, BIG><<<grid, BLOCK, 0, st>>>(P); return 0;
    default: break;
    }
    switch (obj_kind) {
    case CGO_OBJ_QUAD_DIAG: return launch_obj<ObjQuadDiag, BIG>(mode, P, grid, st);
    case CGO_OBJ_ROSENBROCK_PAIRED: return launch_obj<ObjRosenPaired, BIG>(mode, P, grid, st);
    case CGO_OBJ_BOOTH: return launch_obj<ObjBooth, BIG>(mode, P, grid, st);
    default: return -2;
    }
}

// launch a run-time compiled kernel taking one by-value parameter struct
static int launch_module(hipFunction_t f, void *params, int grid, hipStream_t st) {
    if (!f) { set_error("internal: kernel missing from the run-time compiled objective module"); return CGO_EINVAL; }
    void *args[] = {params};
    HIPCHK(hipModuleLaunchKernel(f, grid, 1, 1, BLOCK, 1, 1, 0, st, args, nullptr));
    return CGO_OK;
}

int launch_fused(HipCtx *ctx, int obj_kind, int mode, const void *kparams, int64_t n, bool timed,
                 const HipObjective *obj, hipEvent_t e0, hipEvent_t e1, double big_forced) {
    if (!e0) { e0 = ctx->ev0; e1 = ctx->ev1; }
    const KParams &P = *(const KParams *)kparams;
    const bool hp = obj && obj->uses_param();
    const bool big = is_big(obj_kind, mode, n, hp, big_forced < 0.0 ? env_big_bytes() : big_forced);
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx->stream;
    if (timed) HIPCHK(hipEventRecord(e0, st));
    int r;
    const bool objective_mode = (mode & (M_TRIAL | M_INIT)) != 0;
    if (obj_kind == CGO_OBJ_USER && objective_mode) {
        if (!obj || !obj->rtc) { set_error("user objective has no compiled module"); return CGO_EINVAL; }
        KParams Pc = P;
        if (int rc = launch_module(obj->rtc->fused(mode, big), &Pc, grid, st)) return rc;
        r = 0;
    } else {
        r = big ? launch_any<true>(obj_kind, mode, P, grid, st) : launch_any<false>(obj_kind, mode, P, grid, st);
    }
    if (timed) HIPCHK(hipEventRecord(e1, st));  // brackets k_fused only, not k_finalize
    if (r == -2) { set_error("objective kind not implemented on the device yet"); return CGO_EINVAL; }
    if (r) { set_error("internal: kernel mode not instantiated"); return CGO_EINVAL; }
    HIPCHK(hipGetLastError());
    const bool has_sums = mode != M_ACCEPT;
    if (has_sums) {
        if (int rc = finalize_rows(ctx, grid, NS)) return rc;
    }
    return CGO_OK;
}

// Local sums (device) → global sums (host), identical on every rank:
// all-gather the NS-double block, then add in rank order.

// Single rank: the ctx's own pinned block.  Shared-memory communicator: this rank's slot of the
// segment, double-buffered on the launch sequence number (a rank can run at most one launch ahead
// of the slowest reader: it publishes launch k+1 only after it has consumed every rank's launch k).
void HipCtx::pub_target(double **out, unsigned long long **seqw) {
    *out = nullptr; *seqw = host_seq;
    pub_checked = false;   // (make_tail sets it for a fused launch)
    if (!host_publish) return;
    if (single()) { *out = host_pinned; return; }
    if (shm()) {
        double *d = comm->shm_slot_dev(rank(), (int)(seq & 1));
        *out = d;
        *seqw = (unsigned long long *)(d + 64);
    }
}

// what the word beside a self-validating block of launch `seq` must read (finish_tail)
static inline unsigned long long block_check(unsigned long long seq, const double *block, int ns) {
    unsigned long long c = tail_check_seq(seq);
    for (int t = 0; t < ns; ++t) {
        unsigned long long b;
        std::memcpy(&b, block + t, 8);
        c += tail_check_term(b, t);
    }
    return c;
}
// One look at a published block: copies it to dst and says whether that copy is launch `want`, complete.
// The two publishing formats describe themselves: a finalize kernel (k_finalize_t, k_finalize_lse, k_publish) stores the
// block, fences at system scope and RELEASES the plain sequence number; a fused launch / k_finalize_one stores values and a
// check word over (seq, values) in any order.  A reader that may meet either (`checked`: any peer's mailbox slot — WHICH
// kernel finished a rank's sums depends on that rank's own row count, so ranks with uneven shards publish different
// formats for the same launch) accepts the plain number first, then the check word; the two cannot be confused (a check
// word equals the small integer `want` with probability 2⁻⁶⁴, and a slot's previous launch was `want − 2`).
static inline bool block_ready(bool checked, unsigned long long *word, unsigned long long want, const double *block, int ns, double *dst) {
    const unsigned long long w = __atomic_load_n(word, __ATOMIC_ACQUIRE);
    if (w == want) {
        std::memcpy(dst, block, sizeof(double) * ns);
        return true;
    }
    if (!checked) return false;
    const volatile double *vb = block;
    for (int t = 0; t < ns; ++t) dst[t] = vb[t];
    return w == block_check(want, dst, ns);
}

// every rank's block of launch `want` from the shared segment, rank-major into h[W][ns].  This rank's own slot is
// awaited first, so that the time spent on the others afterwards is what the peers (skew + PCIe latency) cost.
static int shm_collect(HipCtx *ctx, unsigned long long want, double *h, int ns) {
    const int W = ctx->world(), me = ctx->rank();
    double t_own = 0.0, t_start = 0.0;
    for (int k = 0; k < W; ++k) {
        const int r = (k == 0) ? me : (k <= me ? k - 1 : k);   // me, 0, 1, …, me−1, me+1, …
        double *slot = ctx->comm->shm_slot_host(r, (int)(want & 1));
        unsigned long long *sq = (unsigned long long *)(slot + 64);
        unsigned long long spins = 0;
        while (!block_ready(true, sq, want, slot, ns, h + (size_t)r * ns)) {   // either format: see block_ready
            __builtin_ia32_pause();
            if ((++spins & 0xFFFFF) == 0) {
                if (r == me) {  // our own slot: is our stream still alive?
                    hipError_t q = hipStreamQuery(ctx->stream);
                    if (q != hipSuccess && q != hipErrorNotReady) {
                        set_error(std::string("HIP error while waiting for a launch: ") + hipGetErrorString(q));
                        return CGO_EHIP;
                    }
                }
                const double t = now_ns();
                if (t_start == 0.0) t_start = t;
                if (t - t_start > wait_timeout_ns()) {  // a peer died or diverged
                    set_error("shared-memory exchange: rank " + std::to_string(r) + " never published launch " + std::to_string(want));
                    return CGO_ECOMM;
                }
            }
        }
        if (k == 0) t_own = now_ns();
    }
    ctx->xch_peer_wait_ns += now_ns() - t_own;
    ctx->xch_count++;
    return CGO_OK;
}

// spin on the sequence word a kernel releases at system scope into pinned memory
static int wait_word(HipCtx *ctx, unsigned long long *word, unsigned long long want);
static int wait_seq(HipCtx *ctx, unsigned long long want) { return wait_word(ctx, ctx->host_seq, want); }
static int wait_word(HipCtx *ctx, unsigned long long *word, unsigned long long want) {
    unsigned long long spins = 0;
    double t_start = 0.0;
    while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != want) {
        __builtin_ia32_pause();
        if ((++spins & 0xFFFFF) == 0) {  // every ~1M spins: has the stream died or drained?
            const double t = now_ns();
            if (t_start == 0.0) t_start = t;
            if (t - t_start > wait_timeout_ns()) {   // e.g. a collective that a dead peer never joins
                set_error("timed out waiting for a launch to publish its sums (CGO_WAIT_TIMEOUT_S)");
                return CGO_ECOMM;
            }
            hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == want) break;
                set_error("publishing kernel completed but its sequence word never became visible");
                return CGO_EHIP;
            }
            if (q != hipErrorNotReady) {
                set_error(std::string("HIP error while waiting for a launch: ") + hipGetErrorString(q));
                return CGO_EHIP;
            }
        }
    }
    return CGO_OK;
}

// the self-validating block of a fused launch: poll until block and word agree on launch `want`
static int wait_checked(HipCtx *ctx, unsigned long long *word, unsigned long long want, const double *block, int ns, double *dst) {
    unsigned long long spins = 0;
    double t_start = 0.0;
    while (!block_ready(true, word, want, block, ns, dst)) {
        __builtin_ia32_pause();
        if ((++spins & 0xFFFFF) == 0) {
            const double t = now_ns();
            if (t_start == 0.0) t_start = t;
            if (t - t_start > wait_timeout_ns()) {
                set_error("timed out waiting for a launch to publish its sums (CGO_WAIT_TIMEOUT_S)");
                return CGO_ECOMM;
            }
            hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {
                if (block_ready(true, word, want, block, ns, dst)) break;
                set_error("launch completed but its published block never validated");
                return CGO_EHIP;
            }
            if (q != hipErrorNotReady) {
                set_error(std::string("HIP error while waiting for a launch: ") + hipGetErrorString(q));
                return CGO_EHIP;
            }
        }
    }
    return CGO_OK;
}

int fetch_sums(HipCtx *ctx, double *sums, int merge, int ns, double *raw) {
    const int W = ctx->world();
    double *h = ctx->host_pinned;
    if (ctx->single() && ctx->host_publish) {
        if (ctx->pub_checked) {
            if (int rc = wait_checked(ctx, ctx->host_seq, ctx->seq, h, ns, sums)) return rc;
        } else {
            if (int rc = wait_seq(ctx, ctx->seq)) return rc;
            std::memcpy(sums, h, sizeof(double) * ns);
        }
        if (raw) std::memcpy(raw, sums, sizeof(double) * ns);
        return CGO_OK;
    }
    if (ctx->single()) {
        HIPCHK(hipMemcpyAsync(h, ctx->out_dev, sizeof(double) * ns, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::memcpy(sums, h, sizeof(double) * ns);
        if (raw) std::memcpy(raw, h, sizeof(double) * ns);
        return CGO_OK;
    }
    static thread_local std::vector<double> shm_block;  // per-launch path: no allocation after the first call
    int dr = -1;
    bool timed_x = false;
    if (ctx->shm() && ctx->host_publish) {
        if (shm_block.size() < (size_t)W * ns) shm_block.resize((size_t)W * ns);
        if (int rc = shm_collect(ctx, ctx->seq, shm_block.data(), ns)) return rc;
        h = shm_block.data();
        dr = 2;  // blocks already on the host
    } else {
        if (int rc = ctx->ensure_gather()) return rc;
        ctx->xch_collect();
        timed_x = !ctx->xev_pending && (ctx->xch_count & 7) == 0;
        if (timed_x) {
            if (!ctx->xev0) { HIPCHK(hipEventCreate(&ctx->xev0)); HIPCHK(hipEventCreate(&ctx->xev1)); }
            HIPCHK(hipEventRecord(ctx->xev0, ctx->stream));
        }
        dr = ctx->comm->allgather_device(ctx->out_dev, ctx->gather_dev, ns, (void *)ctx->stream);
        ctx->xch_count++;
    }
    if (dr == 2) {
    } else if (dr == 0 && ctx->host_publish) {
        ctx->seq++;
        k_publish<<<1, 64, 0, ctx->stream>>>(ctx->gather_dev, ns * W, h, ctx->host_seq, ctx->seq);
        HIPCHK(hipGetLastError());
        if (timed_x) { HIPCHK(hipEventRecord(ctx->xev1, ctx->stream)); ctx->xev_pending = true; }
        if (int rc = wait_seq(ctx, ctx->seq)) return rc;
    } else if (dr == 0) {
        HIPCHK(hipMemcpyAsync(h, ctx->gather_dev, sizeof(double) * ns * W, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    } else if (dr < 0) {  // host communicator (callback)
        double local[NG];
        HIPCHK(hipMemcpyAsync(h, ctx->out_dev, sizeof(double) * ns, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::memcpy(local, h, sizeof(double) * ns);
        if (ctx->comm->allgather_host(local, h, ns) != 0) {
            set_error("allgather callback failed");
            return CGO_ECOMM;
        }
    } else {
        return CGO_ECOMM;
    }
    if (raw) std::memcpy(raw, h, sizeof(double) * (size_t)ns * W);
    for (int s = 0; s < ns; ++s) {
        double t = 0.0;
        for (int r = 0; r < W; ++r) t += h[r * ns + s];
        sums[s] = t;
    }
    if (merge == MERGE_MAX0) {  // slot 0 is a maximum over ranks
        double m = h[0];
        for (int r = 1; r < W; ++r) if (h[r * ns] > m) m = h[r * ns];
        sums[0] = m;
    }
    if (merge == MERGE_LSE) {  // (max, Σe, Σe·u) merge in rank order instead of plain sums
        double m = h[L_M], S = h[L_S], T = h[L_T];
        for (int r = 1; r < W; ++r) lse_merge(m, S, T, h[r * ns + L_M], h[r * ns + L_S], h[r * ns + L_T]);
        sums[L_M] = m; sums[L_S] = S; sums[L_T] = T;
    }
    return CGO_OK;
}

// Rows of ctx->partials → ctx->out_dev (+ pinned publish).  Two stages once the row block is
// larger than one CU streams in a few µs.
// One workgroup streams ≈ 21–25 GB/s (measured), a dependent second launch costs ≈ 4–5 µs: two stages
// pay off above ≈ 128 KB of rows (256 rows × 56 slots = 112 KB stays single-stage: ≈ 5 µs vs ≈ 9 µs).
static inline bool two_stage_rows(long long rows, int ns) {
    static const long long thr = [] { const char *e = getenv("CGO_FINALIZE_2STAGE_BYTES"); long long v = e ? atoll(e) : 0; return v > 0 ? v : 131072LL; }();
    return rows * ns * 8 > thr;
}

// `canon`: the k_cg / k_chain family.  Its sums have ONE summation order whoever does the summing — these launches or
// the launch's own last workgroup (finish_tail): groups of 64 rows as soon as there are more than 64, interleave
// G = BLOCK / N.  (The stored-gradient family keeps the faster 768-lane single stage up to 128 KB of rows.)
int finalize_rows(HipCtx *ctx, int rows, int ns, bool canon) {
    hipStream_t st = ctx->stream;
    ctx->seq++;
    double *hp; unsigned long long *hs;
    ctx->pub_target(&hp, &hs);
    // two stages in ONE launch (k_finalize_one) wherever two are needed and the mailbox-slot machinery is on
    if (ctx->fused_tail && rows > TAIL_GROUP && rows <= TAIL_GROUP * TAIL_GROUP && (canon || two_stage_rows(rows, ns))) {
        Tail t{};
        t.partials2 = ctx->partials2_f; t.tickets = ctx->tickets; t.out = ctx->out_dev;
        t.host_out = hp; t.host_seq = hs; t.seq = ctx->seq; t.strict = ctx->tail_strict ? 1 : 0;
        ctx->pub_checked = (hp != nullptr) && !ctx->tail_strict;
        const int nb = (rows + TAIL_GROUP - 1) / TAIL_GROUP;
        if (ns == NG) k_finalize_one<NG><<<nb, BLOCK, 0, st>>>(ctx->partials, rows, t);
        else if (ns == NR) k_finalize_one<NR><<<nb, BLOCK, 0, st>>>(ctx->partials, rows, t);
        else if (ns == NR5) k_finalize_one<NR5><<<nb, BLOCK, 0, st>>>(ctx->partials, rows, t);
        else if (ns == NR7) k_finalize_one<NR7><<<nb, BLOCK, 0, st>>>(ctx->partials, rows, t);
        else if (ns == NRC3) k_finalize_one<NRC3><<<nb, BLOCK, 0, st>>>(ctx->partials, rows, t);
        else if (ns == NS) k_finalize_one<NS><<<nb, BLOCK, 0, st>>>(ctx->partials, rows, t);
        else { set_error("internal: no single-launch reduction for this row width"); return CGO_EINVAL; }
        HIPCHK(hipGetLastError());
        return CGO_OK;
    }
    const double *src = ctx->partials;
    int nrows = rows;
    if (canon) {
        if (rows > TAIL_GROUP * TAIL_GROUP) { set_error("internal: more partial rows than two levels of 64 reduce"); return CGO_EINVAL; }
        if (rows > TAIL_GROUP) {
            const int nb = (rows + TAIL_GROUP - 1) / TAIL_GROUP;
            if (ns == NR) k_finalize_t<NR, BLOCK><<<nb, BLOCK, 0, st>>>(ctx->partials, TAIL_GROUP, rows, ctx->partials2, nullptr, nullptr, 0);
            else if (ns == NR5) k_finalize_t<NR5, BLOCK><<<nb, BLOCK, 0, st>>>(ctx->partials, TAIL_GROUP, rows, ctx->partials2, nullptr, nullptr, 0);
            else if (ns == NR7) k_finalize_t<NR7, BLOCK><<<nb, BLOCK, 0, st>>>(ctx->partials, TAIL_GROUP, rows, ctx->partials2, nullptr, nullptr, 0);
            else if (ns == NRC3) k_finalize_t<NRC3, BLOCK><<<nb, BLOCK, 0, st>>>(ctx->partials, TAIL_GROUP, rows, ctx->partials2, nullptr, nullptr, 0);
            else k_finalize_t<NS, BLOCK><<<nb, BLOCK, 0, st>>>(ctx->partials, TAIL_GROUP, rows, ctx->partials2, nullptr, nullptr, 0);
            HIPCHK(hipGetLastError());
            src = ctx->partials2;
            nrows = nb;
        }
        if (ns == NR) k_finalize_t<NR, BLOCK><<<1, BLOCK, 0, st>>>(src, nrows, nrows, ctx->out_dev, hp, hs, ctx->seq);
        else if (ns == NR5) k_finalize_t<NR5, BLOCK><<<1, BLOCK, 0, st>>>(src, nrows, nrows, ctx->out_dev, hp, hs, ctx->seq);
        else if (ns == NR7) k_finalize_t<NR7, BLOCK><<<1, BLOCK, 0, st>>>(src, nrows, nrows, ctx->out_dev, hp, hs, ctx->seq);
        else if (ns == NRC3) k_finalize_t<NRC3, BLOCK><<<1, BLOCK, 0, st>>>(src, nrows, nrows, ctx->out_dev, hp, hs, ctx->seq);
        else k_finalize_t<NS, BLOCK><<<1, BLOCK, 0, st>>>(src, nrows, nrows, ctx->out_dev, hp, hs, ctx->seq);
        HIPCHK(hipGetLastError());
        return CGO_OK;
    }
    if (two_stage_rows(rows, ns)) {
        const int nb = (rows + 63) / 64;
        if (ns == NG) k_finalize_t<NG, 768><<<nb, 768, 0, st>>>(ctx->partials, 64, rows, ctx->partials2, nullptr, nullptr, 0);
        else k_finalize_t<NS, BLOCK><<<nb, BLOCK, 0, st>>>(ctx->partials, 64, rows, ctx->partials2, nullptr, nullptr, 0);
        HIPCHK(hipGetLastError());
        src = ctx->partials2;
        nrows = nb;
    }
    if (ns == NG) k_finalize_t<NG, 768><<<1, 768, 0, st>>>(src, nrows, nrows, ctx->out_dev, hp, hs, ctx->seq);
    else k_finalize_t<NS, BLOCK><<<1, BLOCK, 0, st>>>(src, nrows, nrows, ctx->out_dev, hp, hs, ctx->seq);
    HIPCHK(hipGetLastError());
    return CGO_OK;
}

int finalize_launch(HipCtx *ctx, int grid, bool lse) {
    if (!lse) return finalize_rows(ctx, grid, NS);
    ctx->seq++;
    double *hp; unsigned long long *hs;
    ctx->pub_target(&hp, &hs);
    if (grid > 1024) {   // two stages: 64 rows per workgroup, then one workgroup over the ≤ 64 merged rows
        const int nb = (grid + 63) / 64;
        k_finalize_lse<<<nb, BLOCK, 0, ctx->stream>>>(ctx->partials, 64, grid, ctx->partials2, nullptr, nullptr, 0);
        k_finalize_lse<<<1, BLOCK, 0, ctx->stream>>>(ctx->partials2, nb, nb, ctx->out_dev, hp, hs, ctx->seq);
    } else {
        k_finalize_lse<<<1, BLOCK, 0, ctx->stream>>>(ctx->partials, grid, grid, ctx->out_dev, hp, hs, ctx->seq);
    }
    HIPCHK(hipGetLastError());
    return CGO_OK;
}

// ---------------------------------------------------------------- backend
HipBackend::HipBackend(HipCtx *ctx, HipObjective *obj) : ctx_(ctx), obj_(obj) {
    cgo_solver_policy_init(&pol_);
    epoch_ = ++ctx->solver_epoch;
}
// The RESOLVED policy of this solver (cgo_capi.hip: explicit argument > context default > CGO_* experiment override >
// library): read wherever the backend used to ask the environment.
void HipBackend::set_policy(const cgo_solver_policy &p) {
    pol_ = p;
    ctl_fused_ = p.controller_fused != 0;
    if (p.fused_tail >= 0) ctx_->fused_tail = p.fused_tail != 0;     // context-wide (the row buffers and tickets are the context's)
    if (p.strict_tail >= 0) ctx_->tail_strict = p.strict_tail != 0;
}
HipBackend::~HipBackend() {
    if (pipe_done_ < pipe_enq_ && ctx_->stream) (void)hipStreamSynchronize(ctx_->stream);  // rounds in flight read ctl_dev_
    if (placed_ && x_.p && u_.p && !ctx_->placed_x.p && !ctx_->placed_u.p) {   // the next solver of this size skips the search
        if (ctx_->stream) (void)hipStreamSynchronize(ctx_->stream);
        std::swap(x_.p, ctx_->placed_x.p); std::swap(x_.n, ctx_->placed_x.n);
        std::swap(u_.p, ctx_->placed_u.p); std::swap(u_.n, ctx_->placed_u.n);
        ctx_->placed_n = obj_->n_local;
        ctx_->placed_first_us = place_first_us_; ctx_->placed_best_us = place_best_us_; ctx_->placed_candidates = place_candidates_;
    }
    for (auto &g : graphs_) if (g.exec) (void)hipGraphExecDestroy((hipGraphExec_t)g.exec);
    if (ctl_dev_) (void)hipFree(ctl_dev_);
    if (ctl_rec_) (void)hipHostFree(ctl_rec_);
    if (ctl_seq_) (void)hipHostFree(ctl_seq_);
    if (qn_alpha_dev_) (void)hipFree(qn_alpha_dev_);
    if (res_state_ || res_xbuf_) { if (ctx_->stream) (void)hipStreamSynchronize(ctx_->stream); }
    if (res_state_) (void)hipHostFree(res_state_);
    if (res_recs_) (void)hipHostFree(res_recs_);
    if (res_log_) (void)hipHostFree(res_log_);
    if (res_done_) (void)hipHostFree(res_done_);
    if (res_xbuf_) (void)hipFree(res_xbuf_);
    if (res_recs_dev_) (void)hipFree(res_recs_dev_);
    if (res_log_dev_) (void)hipFree(res_log_dev_);
    if (res_err_) (void)hipFree(res_err_);
    for (auto &r : ring_) { if (r.e0) (void)hipEventDestroy(r.e0); if (r.e1) (void)hipEventDestroy(r.e1); }
}

int HipBackend::alloc() {
    HIPCHK(hipSetDevice(ctx_->device));
    const size_t n = (size_t)obj_->n_local + (chain() ? (size_t)(obj_->n_local & 1) : 0);   // stencil, odd length: one phantom element of padding
    if (ctx_->placed_n != obj_->n_local && (ctx_->placed_x.p || ctx_->placed_u.p)) {   // parked buffers of another size: give them back first (peak memory)
        ctx_->placed_x.release(); ctx_->placed_u.release(); ctx_->placed_n = 0;
    }
    if (int rc = x_.alloc(n)) return rc;
    if (int rc = u_.alloc(n)) return rc;
    if (chain()) {   // the padding (and everything else) starts at zero; the launches keep it there
        HIPCHK(hipMemsetAsync(x_.p, 0, n * sizeof(double), ctx_->stream));
        HIPCHK(hipMemsetAsync(u_.p, 0, n * sizeof(double), ctx_->stream));
    }
    xc_ = x_.p; uc_ = u_.p;   // (place() may swap other buffers in, once the launch policy is known)
    if (chain()) {   // stencil objective: x / u are never updated in place
        if (int rc = x2_.alloc(n)) return rc;
        if (int rc = u2_.alloc(n)) return rc;
        HIPCHK(hipMemsetAsync(x2_.p, 0, n * sizeof(double), ctx_->stream));
        HIPCHK(hipMemsetAsync(u2_.p, 0, n * sizeof(double), ctx_->stream));
        xalt_ = x2_.p; ualt_ = u2_.p;
        pingpong_ = 1;
    }
    // The gradient-free family keeps 16 B/element (+ 8 for a parameter vector) resident: n up to ≈ 1.1e10 in
    // 288 GB.  Its two optional buffers appear on first use: ga_ when a gradient is materialised (results,
    // scaled-norm rare path), gb_ as solvesystem's second iterate.  The stored-gradient families need both now.
    if (!rmode_) {
        if (int rc = ensure_ga()) return rc;
        if (int rc = ensure_gb()) return rc;
    }
    return CGO_OK;
}

// BIG launches of the k_cg family can write x / u out of place when a second pair of buffers fits beside the state.
// OFF unless CGO_PINGPONG=1: the no-arithmetic harness showed out-of-place 10 % ahead on one MI355X (650 vs 720 µs)
// and level on another (717 vs 719 µs), and the engine's own launch gained nothing on either (697 vs 719, 681 vs
// 683 µs; gpurun_out/r02_ab, r02_misc) — not worth 16 B/element of HBM.  Decided once, at the first such launch.
bool HipBackend::pingpong_ready() {
    if (pingpong_ >= 0) return pingpong_ == 1;
    pingpong_ = 0;
    const char *e = getenv("CGO_PINGPONG");
    if (!e || e[0] != '1') return false;
    if (!rmode_ || sys_on_) return false;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
    const size_t need = 2 * sizeof(double) * (size_t)obj_->n_local;
    if (fr < need + (size_t(2) << 30)) return false;   // keep 2 GiB of headroom for ga_/gb_ and the caller
    if (x2_.alloc((size_t)obj_->n_local) != CGO_OK) { (void)hipGetLastError(); return false; }
    if (u2_.alloc((size_t)obj_->n_local) != CGO_OK) { (void)hipGetLastError(); x2_.release(); return false; }
    xalt_ = (xc_ == x_.p) ? x2_.p : x_.p;
    ualt_ = (uc_ == u_.p) ? u2_.p : u_.p;
    pingpong_ = 1;
    return true;
}

// WHERE x, u and D live decides how fast the accept+dir+trial mix runs.  On every MI355X box sampled (five), the same
// no-arithmetic kernel (k_stream_mix: R x,u,D / W x,u in place, n = 1e8) takes ≈ 635 µs on some triples of separately
// allocated buffers and 740–770 µs on others — stable per triple, three levels (≈ 640 / 715 / 755), no rule in the
// virtual addresses, a spacer between the allocations does not help (scripts/tune/rw_mix.hip "place", "spacer", "arena";
// profiles/r02_placement_*.log): DRAM channel/bank conflicts between the physical pages the allocator happened to hand
// out.  The buffers a solver gets by plain consecutive hipMallocs are usually a slow triple (engine launch 750–775 µs).
// So for pure-HBM problem sizes the solver allocates a few spare buffers, times the bare mix on the ordered pairs
// (x, u) of the pool with D where it is, then on the best pair with D moved into each remaining buffer, keeps the
// fastest triple (D is copied once, device to device) and frees the rest: ≈ 80 ms once per solver at n = 1e8, paid back
// within a few hundred iterations.  CGO_PLACE_TUNE=0 switches it off; skipped when the spare buffers do not fit.
int HipBackend::tune_placement() {
    const bool on = pol_.placement_search != 0;
    const int64_t n = obj_->n_local;
    const bool hp = obj_->uses_param();
    static const bool dbg = getenv("CGO_DEBUG_PLACE") != nullptr;
    if (dbg) fprintf(stderr, "[cgo place] on=%d rmode=%d chain=%d bytes=%.3g big=%.3g\n", (int)on, (int)rmode_, (int)chain(),
                     bytes_r(obj_->kind, R_ACCEPT | R_DIR | R_TRIAL, n, hp), big_bytes(false));
    // Searched for the pure-HBM (BIG) launches only.  Round 3 measured the grid-stride launches that exceed the 256 MiB
    // Infinity Cache as well (CGO_PLACE_MIN_BYTES=2.7e8; the 8-GPU shard of config 5, n/8 = 1.25e7, and config 3 at
    // n = 1e7): there the bare mix on the launch's own policy differs by 3–5 % between triples, with no
    // "level" among 134–192 candidates (65.2 → 61.8 µs, 67.5 → 64.3 µs; config 3: 40.4 → 38.9 µs), and the engine's launch
    // does not move at all (87.0 vs 87.2 µs, 53.7 vs 54.6 µs; scripts/r03_shard.sh) — at those sizes the launch is 20 µs above
    // its own mix for other reasons (two waves per SIMD do not hide the FP64 work behind the stream).  Not worth 24 spare
    // buffers and 30–40 ms per solver: off by default below the BIG threshold.
    const double launch_bytes = bytes_r(obj_->kind, R_ACCEPT | R_DIR | R_TRIAL, n, hp);
    static const double min_env = [] { const char *e = getenv("CGO_PLACE_MIN_BYTES"); double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 0.0; }();   // (experiments: the search below the pure-HBM threshold)
    const double min_bytes = min_env > 0.0 ? min_env : big_bytes(false);
    const bool big = launch_bytes > big_bytes(false);
    const int mix_grid = big ? GRID_BIG : grid_cg(n, policy_points());
    if (!on || !rmode_ || chain() || launch_bytes <= min_bytes) return CGO_OK;
    HIPCHK(hipSetDevice(ctx_->device));
    if (ctx_->placed_n == n && ctx_->placed_x.p && ctx_->placed_u.p) {   // an earlier solver of this size already searched
        x_.release(); u_.release();
        std::swap(x_.p, ctx_->placed_x.p); std::swap(x_.n, ctx_->placed_x.n);
        std::swap(u_.p, ctx_->placed_u.p); std::swap(u_.n, ctx_->placed_u.n);
        place_first_us_ = ctx_->placed_first_us; place_best_us_ = ctx_->placed_best_us; place_candidates_ = ctx_->placed_candidates;
        placed_ = true;
        return CGO_OK;
    }
    if (ctx_->placed_x.p || ctx_->placed_u.p) {   // parked buffers of another size: give them back before searching
        ctx_->placed_x.release(); ctx_->placed_u.release(); ctx_->placed_n = 0;
    }
    // Spare buffers come in stages of eight, up to three stages (CGO_PLACE_STAGES): whether a process's allocations hold a
    // fast triple at all is a matter of luck — on one box three processes of four found none among 64 candidates from
    // x, u, D + 8 spares, the fourth at its 8th candidate (gpurun_out/r02_fin1) — and new allocations made while the old
    // ones are held land on other physical pages.
    constexpr int STAGE = 8, SPARE = 3 * STAGE, PER_STAGE = 64;
    const int stages = (pol_.placement_stages >= 1 && pol_.placement_stages <= 3) ? pol_.placement_stages : 3;
    const size_t vec = (size_t)n * sizeof(double);
    hipStream_t st = ctx_->stream;
    DevBuf spare[SPARE];
    int have = 0;
    std::vector<double *> pool = {x_.p, u_.p};
    // The search's TRANSIENT memory is capped: policy.placement_max_bytes, or — library policy — a quarter of what is free now
    // (never more than the 24 vectors of three stages).  Below one stage's worth it does not run.
    int max_spares = SPARE;
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return CGO_OK; }
        const double cap = pol_.placement_max_bytes > 0 ? (double)pol_.placement_max_bytes : 0.25 * (double)fr;
        max_spares = (int)std::min<double>((double)SPARE, cap / (double)vec);
        place_cap_bytes_ = (double)max_spares * (double)vec;
        if (max_spares < STAGE) { if (dbg) fprintf(stderr, "[cgo place] memory cap %.3g B < one stage of spares: no search\n", cap); return CGO_OK; }
    }
    auto grow = [&]() -> int {   // one more stage of spares, as far as memory allows (ga_/gb_ and the caller need room too)
        int added = 0;
        while (have < SPARE && have < max_spares && added < STAGE) {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); break; }
            if (fr < 2 * vec + (size_t(4) << 30)) break;
            if (spare[have].alloc((size_t)n) != CGO_OK) { (void)hipGetLastError(); break; }
            if (hipMemsetAsync(spare[have].p, 0, vec, st) != hipSuccess) { (void)hipGetLastError(); break; }
            pool.push_back(spare[have].p);
            ++have; ++added;
        }
        if (dbg) fprintf(stderr, "[cgo place] +%d spare buffers (%d in the pool)\n", added, (int)pool.size());
        return added;
    };
    if (grow() < 2) return CGO_OK;
    HIPCHK(hipMemsetAsync(x_.p, 0, vec, st));
    HIPCHK(hipMemsetAsync(u_.p, 0, vec, st));
    auto time_mix = [&](double *x, double *u, const double *d, double &us) -> int {
        float t[2];
        for (int r = -1; r < 2; ++r) {
            if (r >= 0) HIPCHK(hipEventRecord(ctx_->ev0, st));
            if (big) {
                if (hp) k_stream_mix<true, true><<<mix_grid, BLOCK, 0, st>>>(x, u, d, n, 1e-9, 0.5);
                else k_stream_mix<false, true><<<mix_grid, BLOCK, 0, st>>>(x, u, d, n, 1e-9, 0.5);
            } else {   // the streaming policy the engine's launch will use at this size
                if (hp) k_stream_mix<true, false><<<mix_grid, BLOCK, 0, st>>>(x, u, d, n, 1e-9, 0.5);
                else k_stream_mix<false, false><<<mix_grid, BLOCK, 0, st>>>(x, u, d, n, 1e-9, 0.5);
            }
            if (r >= 0) {
                HIPCHK(hipEventRecord(ctx_->ev1, st));
                HIPCHK(hipStreamSynchronize(st));
                HIPCHK(hipEventElapsedTime(&t[r], ctx_->ev0, ctx_->ev1));
            }
        }
        HIPCHK(hipGetLastError());
        us = (double)std::min(t[0], t[1]) * 1e3;
        return CGO_OK;
    };
    const double *d0 = hp ? obj_->p0.p : nullptr;
    double best = 0.0, first = 0.0, worst = 0.0;
    int bx = 0, bu = 1, bd = -1;   // bd = −1: D stays where it is
    // The times come in levels ≈ 10–15 % apart (≈ 640 / 715 / 755 µs at n = 1e8 — none, one, several of the three streams
    // in conflict): stop as soon as a triple sits a level below the slowest seen.  Triples (x, u, D) are drawn from the pool
    // in a fixed pseudo-random order (D may stay where it is or move into a pool buffer); at most 64 per stage are timed.
    if (int rc = time_mix(pool[0], pool[1], d0, first)) return rc;
    best = worst = first; place_candidates_ = 1;
    unsigned long long lcg = 0x9E3779B97F4A7C15ull;
    auto next = [&](int m) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (int)((lcg >> 33) % (unsigned)m); };
    static const double ratio = [] { const char *e = getenv("CGO_PLACE_RATIO"); double v = e ? atof(e) : 0.0; return (v > 0.0 && v < 1.0) ? v : 0.88; }();
    auto found = [&] { return place_candidates_ >= 4 && best <= ratio * worst; };
    // Round 3: "a level below the slowest seen" used to end the search at the MIDDLE level too (667–670 µs at n = 1e8: 3 of 8
    // fresh processes in profiles/r03_headline_samples.txt stopped there after 5–65 candidates, 4 reached 643–646 µs).  The
    // levels are physical — 6.2 / 6.0 / 5.3 TB/s of the five-stream mix on every box sampled — so the top one has an absolute
    // mark: inside a stage the search now goes on until a triple streams at ≥ 6.1 TB/s (or the stage's 64 candidates are
    // used up: ≈ 50 ms at n = 1e8); further stages of spares are still added only while not even the middle level is in hand.
    static const double fast_tbps = [] { const char *e = getenv("CGO_PLACE_FAST_TBPS"); double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 6.1; }();
    const double fast_us = (big && hp) ? 40.0 * (double)n / (fast_tbps * 1e12) * 1e6 : 0.0;
    auto done = [&] { return fast_us > 0.0 ? (place_candidates_ >= 2 && best <= fast_us) : found(); };
    for (int stage = 0; stage < stages && !found() && !done(); ++stage) {
        if (stage > 0 && grow() == 0) break;
        const int P = (int)pool.size();
        for (int it = 0; it < PER_STAGE - (stage == 0 ? 1 : 0) && !done(); ++it) {
            const int i = next(P);
            int j = next(P - 1); if (j >= i) ++j;
            int k = -1;
            // (D moves only while this solver is the objective's only user)
            if (hp && obj_->users <= 1 && next(4) != 0) { k = next(P - 2); const int lo = std::min(i, j), hi2 = std::max(i, j); if (k >= lo) ++k; if (k >= hi2) ++k; }
            double us = 0.0;
            if (int rc = time_mix(pool[i], pool[j], k >= 0 ? pool[k] : d0, us)) return rc;
            place_candidates_++;
            if (us < best) { best = us; bx = i; bu = j; bd = k; }
            if (us > worst) worst = us;
        }
    }
    place_first_us_ = first; place_best_us_ = best;
    if (dbg) fprintf(stderr, "[cgo place] %d candidates: as allocated %.1f us, best %.1f us (x=%d u=%d d=%d)\n", place_candidates_, first, best, bx, bu, bd);
    // hand the chosen buffers to x_, u_ (and the objective's parameter vector); everything else is released
    auto owner = [&](double *p) -> DevBuf * {
        if (p == x_.p) return &x_;
        if (p == u_.p) return &u_;
        for (auto &sb : spare) if (sb.p == p) return &sb;
        return nullptr;
    };
    double *px = pool[bx], *pu = pool[bu], *pd = bd >= 0 ? pool[bd] : nullptr;
    if (pd) {
        HIPCHK(hipMemcpyAsync(pd, obj_->p0.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        DevBuf *o = owner(pd);
        std::swap(o->p, obj_->p0.p); std::swap(o->n, obj_->p0.n);
    }
    if (px != x_.p) { DevBuf *o = owner(px); std::swap(o->p, x_.p); std::swap(o->n, x_.n); }
    if (pu != u_.p) { DevBuf *o = owner(pu); std::swap(o->p, u_.p); std::swap(o->n, u_.n); }
    HIPCHK(hipStreamSynchronize(st));
    placed_ = true;
    return CGO_OK;   // the spare DevBufs (now holding the rejected buffers) free themselves here
}

// The placement search, after the C API has set the launch policy (its stream mix runs on the grid the solver's launches will use).
int HipBackend::place() {
    if (int rc = tune_placement()) return rc;
    xc_ = x_.p; uc_ = u_.p;
    return CGO_OK;
}

int HipBackend::ensure_ga() {
    if (!ga_.p) { if (int rc = ga_.alloc((size_t)obj_->n_local + (size_t)(obj_->n_local & 1))) return rc; }
    if (!g_) g_ = ga_.p;
    return CGO_OK;
}
int HipBackend::ensure_gb() {
    if (!gb_.p) { if (int rc = gb_.alloc((size_t)obj_->n_local)) return rc; }
    if (!gt_) gt_ = gb_.p;
    xn_ = (xc_ == gb_.p) ? x_.p : gb_.p;
    return CGO_OK;
}

int HipBackend::set_x0_host(const double *x0) {
    if (int rc = pipe_drain()) return rc;
    discard_pending();   // a new x_initial: nothing of an earlier solve may be applied to it
    HIPCHK(hipSetDevice(ctx_->device));
    xc_ = x_.p; xn_ = gb_.p;   // gb_ may not exist yet (sys_begin creates it)
    if (pingpong_ == 1) xalt_ = x2_.p;
    HIPCHK(hipMemcpyAsync(xc_, x0, sizeof(double) * (size_t)obj_->n_local, hipMemcpyHostToDevice, ctx_->stream));
    HIPCHK(hipStreamSynchronize(ctx_->stream));
    return CGO_OK;
}

int HipBackend::set_x0_device(const double *x0_dev) {
    if (int rc = pipe_drain()) return rc;
    discard_pending();   // a new x_initial: nothing of an earlier solve may be applied to it
    HIPCHK(hipSetDevice(ctx_->device));
    xc_ = x_.p; xn_ = gb_.p;
    if (pingpong_ == 1) xalt_ = x2_.p;
    HIPCHK(hipMemcpyAsync(xc_, x0_dev, sizeof(double) * (size_t)obj_->n_local, hipMemcpyDeviceToDevice, ctx_->stream));
    HIPCHK(hipStreamSynchronize(ctx_->stream));   // the caller may reuse its buffer as soon as this returns
    return CGO_OK;
}

int fill_device(HipCtx *ctx, double *v, int64_t n, int64_t offset, int kind, uint64_t seed, double lo,
                double hi) {
    HIPCHK(hipSetDevice(ctx->device));
    int grid = (int)std::min<int64_t>((n + BLOCK - 1) / BLOCK, GRID_SMALL);
    if (grid < 1) grid = 1;
    k_fill<<<grid, BLOCK, 0, ctx->stream>>>(v, n, offset, kind, seed, lo, hi);
    HIPCHK(hipGetLastError());
    return CGO_OK;
}

int HipBackend::set_x0_fill(int kind, uint64_t seed, double lo, double hi) {
    if (int rc = pipe_drain()) return rc;
    discard_pending();   // a new x_initial: nothing of an earlier solve may be applied to it
    xc_ = x_.p; xn_ = gb_.p;
    if (pingpong_ == 1) xalt_ = x2_.p;
    return fill_device(ctx_, xc_, obj_->n_local, obj_->offset, kind, seed, lo, hi);
}

// Profiling without perturbing the timed region: a launch gets its own pair of HIP events from a
// ring, recorded on the ctx stream around the kernel (not the finalize); elapsed times are read only
// when the ring fills up or the totals are asked for — no per-launch synchronise.  Recording two
// events costs ≈ 4 µs of host time per launch — 16 % of an iteration at n = 1e6 (33.5k vs 40.0k it/s) —
// and still 5–7 % at n = 1e7 — so below n_local = 3e7 only every 4th launch is timed; every launch is COUNTED, and the reported
// time of a kernel kind is (mean of its timed launches) × (its launch count).
int HipBackend::prof_slot(hipEvent_t *e0, hipEvent_t *e1) {
    if (ring_.empty()) {
        ring_.resize(1024);
        for (auto &r : ring_) { HIPCHK(hipEventCreate(&r.e0)); HIPCHK(hipEventCreate(&r.e1)); r.kk = -1; }
    }
    if (ring_used_ == (int)ring_.size()) prof_flush();
    ProfSlot &r = ring_[ring_used_++];
    r.kk = -1;
    *e0 = r.e0; *e1 = r.e1;
    return CGO_OK;
}
// The ring of event pairs is created HERE, not at the first timed launch: 2 048 hipEventCreate calls cost ≈ 200 µs, which used
// to land inside the first timed window (invisible beside 50 launches of 700 µs, most of a 15-iteration resident slice).
void HipBackend::profile_enable(bool on) {
    prof_on_ = on;
    if (on && ring_.empty() && hipSetDevice(ctx_->device) == hipSuccess) {
        ring_.resize(1024);
        for (auto &r : ring_) { if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { (void)hipGetLastError(); } r.kk = -1; }
    }
}

bool HipBackend::prof_pick(int kk) {  // per kernel kind, so the first launch of every kind is timed
    const int every = obj_->n_local >= 30000000 ? 1 : prof_every_;
    if (capturing_) { prof_cur_ = false; return false; }   // no event records inside a stream capture
    prof_cur_ = prof_on_ && (prof_tick_[kk]++ % every) == 0;
    return prof_cur_;
}
int HipBackend::prof_begin(int kk) {
    if (!prof_pick(kk)) return CGO_OK;
    hipEvent_t e0, e1;
    if (int rc = prof_slot(&e0, &e1)) return rc;
    HIPCHK(hipEventRecord(e0, ctx_->stream));
    return CGO_OK;
}
int HipBackend::prof_end() {
    if (!prof_cur_ || ring_used_ == 0) return CGO_OK;
    HIPCHK(hipEventRecord(ring_[ring_used_ - 1].e1, ctx_->stream));
    return CGO_OK;
}
void HipBackend::prof_commit(int kk, double bytes) {
    prof_cnt_[kk]++;
    prof_bytes_[kk] = bytes;
    if (!prof_cur_ || ring_used_ == 0) return;
    ring_[ring_used_ - 1].kk = kk;
    ring_[ring_used_ - 1].bytes = bytes;
    prof_cur_ = false;
}
void HipBackend::prof_flush() {
    if (ring_used_ == 0) return;
    (void)hipStreamSynchronize(ctx_->stream);
    for (int i = 0; i < ring_used_; ++i) {
        ProfSlot &r = ring_[i];
        float ms = 0;
        if (r.kk >= 0 && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            prof_n_[r.kk]++; prof_ms_[r.kk] += ms; prof_bytes_[r.kk] = r.bytes;
        }
    }
    ring_used_ = 0;
    prof_gen_++;
}
void HipBackend::profile_reset() {
    prof_flush();
    for (int k = 0; k < KK_COUNT; ++k) { prof_n_[k] = 0; prof_ms_[k] = 0; prof_bytes_[k] = 0; prof_cnt_[k] = 0; }
    for (int k = 0; k < KK_COUNT; ++k) prof_tick_[k] = 0;
}
void HipBackend::profile_get(int kind, int64_t *launches, double *ms, double *bytes) {
    prof_flush();
    if (kind < 0 || kind >= KK_COUNT) { *launches = 0; *ms = 0; *bytes = 0; return; }
    *launches = prof_cnt_[kind];
    *ms = prof_n_[kind] ? prof_ms_[kind] / (double)prof_n_[kind] * (double)prof_cnt_[kind] : 0.0;
    *bytes = prof_bytes_[kind];
}

int HipBackend::launch(int kk, int mode, double a_acc, double beta, double a_trial, bool fetch,
                       double *sums) {
    HIPCHK(hipSetDevice(ctx_->device));
    if (obj_->uses_param() && !obj_->p0_set && (mode & (M_TRIAL | M_INIT))) {
        set_error("objective parameter vector (slot 0) was never set");
        return CGO_ESTATE;
    }
    KParams P;
    P.x = xc_; P.u = u_.p; P.g = g_; P.gt = gt_; P.p0 = obj_->p0.p;
    P.n = obj_->n_local; P.offset = obj_->offset;
    P.a_acc = a_acc; P.beta = beta; P.a_trial = a_trial; P.s0 = obj_->s0;
    P.partials = ctx_->partials; P.out = ctx_->out_dev;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = prof_pick(kk);
    if (timed) { if (int rc = prof_slot(&e0, &e1)) return rc; }
    if (int rc = launch_fused(ctx_, obj_->kind, mode, &P, obj_->n_local, timed, obj_, e0, e1, pol_.hbm_stream_bytes)) return rc;
    total_launches_++;
    if (fetch) {
        if (int rc = fetch_sums(ctx_, sums)) return rc;
    }
    if (prof_on_) prof_commit(kk, bytes_for(obj_->kind, mode, obj_->n_local, obj_->uses_param()));
    return CGO_OK;
}

static void unpack(const double *s, Scal &o, bool trial, bool dir) {
    if (trial) {
        o.f = s[S_F]; o.gtu = s[S_GTU]; o.gtgt = s[S_GTGT]; o.gtg = s[S_GTG];
        o.yy = s[S_YY]; o.uy = s[S_UY]; o.ygt = s[S_YGT];
    }
    if (dir) { o.gu = s[S_GU]; o.uu = s[S_UU]; }
}

int HipBackend::init_eval(Scal &out) {
    if (int rc = flush_lite()) return rc;
    if (rmode_) {
        double s[NR7];
        if (chain() && !ctx_->single()) {   // the neighbours' edge elements of x0, before the first gradient
            if (int rc = launch_r(KK_INIT, R_EDGES, 0, 0, nullptr, 0, true, s)) return rc;
        }
        if (int rc = launch_r(KK_INIT, R_INIT, 0, 0, nullptr, 0, true, s)) return rc;
        out = Scal();
        out.f = s[RS_F]; out.gtgt = s[RS_GTGT];
        return CGO_OK;
    }
    if (obj_->host_closure()) return host_trial(0.0, true, out);
    if (obj_->two_phase()) {
        out = Scal();
        if (int rc = lse_stats(LM_NOU, 0, 0, 0, out, false)) return rc;
        const double f = out.f;
        if (int rc = lse_grad(true, 0.0, out)) return rc;
        out.f = f;
        std::swap(g_, gt_);
        return CGO_OK;
    }
    double s[NS];
    if (int rc = launch(KK_INIT, M_INIT, 0, 0, 0, true, s)) return rc;
    std::swap(g_, gt_);  // the gradient just written becomes the current one
    out = Scal();
    out.f = s[S_F];
    out.gtgt = s[S_GTGT];
    return CGO_OK;
}

int HipBackend::trial(const double *a, int k, Scal *out) {
    if (rmode_) {
        double s[NR7];
        if (int rc = launch_r(KK_TRIAL, R_TRIAL, 0, 0, a, k, true, s)) return rc;
        unpack_r(s, k, out, false);
        return CGO_OK;
    }
    if (obj_->host_closure()) return host_trial(a[0], false, out[0]);
    if (obj_->two_phase()) return lse_stats(0, 0, 0, a[0], out[0], false);
    if (int rc = flush_lite()) return rc;
    spec_valid_ = false; spec_unmat_ = false;   // this launch writes g⁺ of ITS step: whatever the direction pass speculated on is no longer the last trial
    double s[NS];
    const int mode = need_beta_ ? (M_TRIAL | M_BETA) : M_TRIAL;
    if (int rc = launch(KK_TRIAL, mode, 0, 0, a[0], true, s)) return rc;
    unpack(s, out[0], true, false);
    return CGO_OK;
}

int HipBackend::accept_dir_trial(double a_acc, double beta, const double *a, int k, Scal *out) {
    if (rmode_) {
        double s[NR7];
        if (int rc = launch_r(KK_ACCEPT_DIR_TRIAL, R_ACCEPT | R_DIR | R_TRIAL, a_acc, beta, a, k, true, s)) return rc;
        unpack_r(s, k, out, true);
        return CGO_OK;
    }
    double s[NS];
    if (obj_->host_closure()) {   // accept + direction on the device, then the first trial of the next search
        Scal d;
        if (int rc = accept_dir(a_acc, beta, d)) return rc;
        if (int rc = host_trial(a[0], false, out[0])) return rc;
        out[0].gu = d.gu; out[0].uu = d.uu;
        return CGO_OK;
    }
    std::swap(g_, gt_);  // g ← g⁺ (optim.jl:139) without moving a byte
    if (obj_->two_phase()) return lse_stats(LM_ACCEPT | LM_DIR, a_acc, beta, a[0], out[0], true);
    if (int rc = launch(KK_ACCEPT_DIR_TRIAL, M_ACCEPT | M_DIR | M_TRIAL | M_BETA, a_acc, beta, a[0], true, s))
        return rc;
    unpack(s, out[0], true, true);
    return CGO_OK;
}

int HipBackend::accept_dir(double a_acc, double beta, Scal &out) {
    if (rmode_) {
        double s[NR7];
        if (int rc = launch_r(KK_ACCEPT_DIR, R_ACCEPT | R_DIR, a_acc, beta, nullptr, 0, true, s)) return rc;
        out.gu = s[RS_PER_POINT]; out.uu = s[RS_PER_POINT + 1];
        return CGO_OK;
    }
    double s[NS];
    std::swap(g_, gt_);
    if (int rc = launch(KK_ACCEPT_DIR, M_ACCEPT | M_DIR, a_acc, beta, 0, true, s)) return rc;
    unpack(s, out, false, true);
    return CGO_OK;
}

int HipBackend::accept_only(double a_acc) {
    if (rmode_) return launch_r(KK_ACCEPT_ONLY, R_ACCEPT, a_acc, 0, nullptr, 0, false, nullptr);
    std::swap(g_, gt_);
    return launch(KK_ACCEPT_ONLY, M_ACCEPT, a_acc, 0, 0, false, nullptr);
}

int HipBackend::reset_dir(Scal &out) {
    if (int rc = flush_lite()) return rc;
    if (rmode_) {
        double s[NR7];
        if (int rc = launch_r(KK_RESET_DIR, R_RESET, 0, 0, nullptr, 0, true, s)) return rc;
        out.gu = s[RS_PER_POINT]; out.uu = s[RS_PER_POINT + 1];
        return CGO_OK;
    }
    double s[NS];
    if (int rc = launch(KK_RESET_DIR, M_RESET, 0, 0, 0, true, s)) return rc;
    unpack(s, out, false, true);
    return CGO_OK;
}

int HipBackend::upg_sumsq(double &out) {
    if (rmode_) {
        double s[NR7];
        if (int rc = launch_r(KK_UPG_NORM, R_UPG, 0, 0, nullptr, 0, true, s)) return rc;
        out = s[RS_PER_POINT + 1];
        return CGO_OK;
    }
    double s[NS];
    if (int rc = launch(KK_UPG_NORM, M_UPG, 0, 0, 0, true, s)) return rc;
    out = s[S_UU];
    return CGO_OK;
}

// ---- solvesystem (solve_system.jl) on the gradient-free family ------------------------------
int HipBackend::sys_begin() {
    if (!rmode_) { set_error("solvesystem needs an element-wise objective (k_cg kernel family)"); return CGO_EINVAL; }
    if (int rc = pipe_drain()) return rc;
    sys_on_ = true;
    HIPCHK(hipSetDevice(ctx_->device));
    if (int rc = ensure_gb()) return rc;
    HIPCHK(hipMemcpyAsync(xn_, xc_, sizeof(double) * (size_t)obj_->n_local, hipMemcpyDeviceToDevice, ctx_->stream));  // :82
    return CGO_OK;
}

int HipBackend::sys_project(double a, double m, Scal &out) {
    double s[NR7];
    const double a1[1] = {a};
    if (int rc = launch_r(KK_SYS_PROJECT, R_PROJ, 0.0, m, a1, 1, true, s)) return rc;
    unpack_r(s, 1, &out, false);
    return CGO_OK;
}

int HipBackend::sys_commit() { std::swap(xc_, xn_); return CGO_OK; }  // x, x_next = x_next, x  (:194)

int HipBackend::dir_trial(double beta, const double *a, int k, Scal *out) {
    double s[NR7];
    if (k <= 0) {
        if (int rc = launch_r(KK_DIR_TRIAL, R_DIR, 0.0, beta, nullptr, 0, true, s)) return rc;
        out[0].gu = s[RS_PER_POINT]; out[0].uu = s[RS_PER_POINT + 1];
        return CGO_OK;
    }
    if (int rc = launch_r(KK_DIR_TRIAL, R_DIR | R_TRIAL, 0.0, beta, a, k, true, s)) return rc;
    unpack_r(s, k, out, true);
    return CGO_OK;
}

// ---- gradient-free multi-point CG family (cgo_kernels_cg.hip.hpp) ---------------------------
static double bytes_r(int obj_kind, int mode, int64_t n, bool has_param) {
    const int p = (obj_kind == CGO_OBJ_QUAD_DIAG || has_param) ? 1 : 0;
    int v = 0;
    if (mode == R_INIT) v = 1 + p + 1;
    else if (mode == R_TRIAL) v = 2 + p;
    else if (mode == (R_ACCEPT | R_DIR | R_TRIAL)) v = 2 + p + 2;
    else if (mode == (R_ACCEPT | R_DIR)) v = 2 + p + 2;
    else if (mode == R_ACCEPT) v = 2 + 1;
    else if (mode == R_RESET) v = 1 + p + 1;
    else if (mode == R_UPG) v = 2 + p;
    else if (mode == R_GRAD) v = 1 + p + 1;
    else if (mode == R_GRADT) v = 2 + p + 1;
    else if (mode == R_DIR || mode == (R_DIR | R_TRIAL)) v = 2 + p + 1;
    else if (mode == R_PROJ) v = 3 + p + 1;
    else if (mode == R_EDGES) v = 0;
    return 8.0 * (double)n * (double)v;
}

template <class Obj, bool BIG>
static int launch_cg(int mode, int npts, const RParams &P, int grid, hipStream_t st) {
    switch (mode) {
    case R_INIT: k_cg<Obj, R_INIT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_TRIAL:
        if (npts == 1) k_cg<Obj, R_TRIAL, 1, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 3) k_cg<Obj, R_TRIAL, 3, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 5) k_cg<Obj, R_TRIAL, 5, BIG><<<grid, BLOCK, 0, st>>>(P);
        else k_cg<Obj, R_TRIAL, 7, BIG><<<grid, BLOCK, 0, st>>>(P);
        break;
    case R_ACCEPT | R_DIR | R_TRIAL:
        if (P.tail.ctl) {   // a whole controller round in this launch (never BIG: pipe_fused)
            if (npts == 1) k_cg_armed<Obj, 1><<<grid, BLOCK, 0, st>>>(P);
            else if (npts == 3) k_cg_armed<Obj, 3><<<grid, BLOCK, 0, st>>>(P);
            else if (npts == 5) k_cg_armed<Obj, 5><<<grid, BLOCK, 0, st>>>(P);
            else k_cg_armed<Obj, 7><<<grid, BLOCK, 0, st>>>(P);
            break;
        }
        if (npts == 1) k_cg<Obj, R_ACCEPT | R_DIR | R_TRIAL, 1, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 3) k_cg<Obj, R_ACCEPT | R_DIR | R_TRIAL, 3, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 5) k_cg<Obj, R_ACCEPT | R_DIR | R_TRIAL, 5, BIG><<<grid, BLOCK, 0, st>>>(P);
        else k_cg<Obj, R_ACCEPT | R_DIR | R_TRIAL, 7, BIG><<<grid, BLOCK, 0, st>>>(P);
        break;
    case R_ACCEPT | R_DIR: k_cg<Obj, R_ACCEPT | R_DIR, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_ACCEPT: k_cg<Obj, R_ACCEPT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_RESET: k_cg<Obj, R_RESET, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_UPG: k_cg<Obj, R_UPG, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_GRAD: k_cg<Obj, R_GRAD, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_GRADT: k_cg<Obj, R_GRADT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_DIR: k_cg<Obj, R_DIR, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_DIR | R_TRIAL:
        if (npts == 1) k_cg<Obj, R_DIR | R_TRIAL, 1, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 3) k_cg<Obj, R_DIR | R_TRIAL, 3, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 5) k_cg<Obj, R_DIR | R_TRIAL, 5, BIG><<<grid, BLOCK, 0, st>>>(P);
        else k_cg<Obj, R_DIR | R_TRIAL, 7, BIG><<<grid, BLOCK, 0, st>>>(P);
        break;
    case R_PROJ: k_cg<Obj, R_PROJ, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    default: return -1;
    }
    return 0;
}

// Row width of a CG launch: 7 sums per trial point + 2 direction sums, padded (10 or 24).
static inline int rows_for(int npts) { return npts == 1 ? NR1 : (npts == 3 ? NR : (npts == 5 ? NR5 : NR7)); }
static inline int npts_for(int k) { return k <= 1 ? 1 : (k <= 3 ? 3 : (k <= 5 ? 5 : 7)); }  // kernel variant for k trial steps

// stencil launches carry one or three trial points: three only where the mode evaluates trials at all
static inline int chain_npts(int mode, int npts) { return ((mode & R_TRIAL) && npts >= 3) ? 3 : 1; }

int HipBackend::launch_r(int kk, int mode, double a_acc, double beta, const double *a, int k, bool fetch,
                         double *sums) {
    if (int rc = pipe_drain()) return rc;
    pipe_streak_ = 0;  // a host-driven launch: the streak of controller-eligible launches ends
    int grid = 0;
    const int npts = npts_for(k);
    if (int rc = launch_r_kernel(kk, mode, a_acc, beta, a, k, npts, nullptr, &grid)) return rc;
    total_launches_++;
    const bool has_sums = (mode != R_ACCEPT && mode != R_GRAD && mode != R_GRADT);
    const bool fused = has_sums && tail_fused(grid);   // the launch's last workgroup already left the sums (finish_tail)
    if (has_sums && chain()) {   // 24- or 32-slot rows: the sums + this rank's eight edge values (cgo_kernels_chain.hip.hpp)
        const bool three = chain_npts(mode, npts) == 3;
        const int W = three ? NRC3 : NRC, edge = three ? RC3_EDGE : RC_EDGE, nsums = three ? NR : NR1;
        if (!fused) { if (int rc = finalize_rows(ctx_, grid, W, true)) return rc; }
        const int Wd = ctx_->world(), me = ctx_->rank();
        std::vector<double> raw((size_t)W * Wd);
        double all[NRC3];
        if (int rc = fetch_sums(ctx_, all, MERGE_SUM, W, raw.data())) return rc;
        if (sums) std::memcpy(sums, all, sizeof(double) * nsums);
        if (me > 0) {           // left neighbour's LAST two elements
            const double *e = raw.data() + (size_t)(me - 1) * W + edge + 4;
            halo_xl_[0] = e[0]; halo_xl_[1] = e[1]; halo_ul_[0] = e[2]; halo_ul_[1] = e[3];
        }
        if (me < Wd - 1) {       // right neighbour's FIRST two elements
            const double *e = raw.data() + (size_t)(me + 1) * W + edge;
            halo_xr_[0] = e[0]; halo_xr_[1] = e[1]; halo_ur_[0] = e[2]; halo_ur_[1] = e[3];
        }
    } else if (has_sums) {
        if (!fused) { if (int rc = finalize_rows(ctx_, grid, rows_for(npts), true)) return rc; }
        if (fetch) {
            if (int rc = fetch_sums(ctx_, sums, MERGE_SUM, rows_for(npts))) return rc;
        }
    }
    if (prof_on_) prof_commit(kk, bytes_r(obj_->kind, mode, obj_->n_local, obj_->uses_param()));
    return CGO_OK;
}

// Fused reduction tail (finish_tail): a host-driven launch of the k_cg / k_chain family takes the next sequence number
// itself and publishes where a finalize launch would have.
// Only where the launch is short: at 4096 workgroups the ≈ 0.5 M slot and ticket atomics and the finisher's chain cost the
// pure-HBM launch what the two finalize launches did (n = 1e8: 671 → 683 µs, 1 236 vs 1 230 it/s; gpurun_out/r02_ft).
// A controller-armed round as ONE launch (tail_ctl): wherever the fused tail applies, except for run-time compiled
// objectives, whose kernels carry no controller code.
bool HipBackend::pipe_fused(int grid) const {
    return tail_fused(grid) && obj_->kind != CGO_OBJ_USER && !chain() && ctl_fused_;
}
bool HipBackend::tail_fused(int grid) const {
    static const int cap = [] { const char *e = getenv("CGO_FUSED_TAIL_MAX_GRID"); int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    return ctx_->fused_tail && grid <= cap && grid <= TAIL_GROUP * TAIL_GROUP;
}
Tail HipBackend::make_tail(bool on) {
    Tail t{};
    if (!on) return t;
    ctx_->seq++;
    t.partials2 = ctx_->partials2_f; t.tickets = ctx_->tickets; t.out = ctx_->out_dev;
    t.strict = ctx_->tail_strict ? 1 : 0;
    ctx_->pub_target(&t.host_out, &t.host_seq);
    ctx_->pub_checked = (t.host_out != nullptr) && !ctx_->tail_strict;
    t.seq = ctx_->seq;
    return t;
}

// the k_cg launch itself (bracketed by the profiling events); `ctl` non-null = controller-armed
int HipBackend::launch_r_kernel(int kk, int mode, double a_acc, double beta, const double *a, int k, int npts,
                                const CtlArgs *ctl, int *grid_out) {
    HIPCHK(hipSetDevice(ctx_->device));
    if (obj_->uses_param() && !obj_->p0_set) { set_error("objective parameter vector (slot 0) was never set"); return CGO_ESTATE; }
    const int64_t n = obj_->n_local;
    if (mode & (R_GRAD | R_GRADT)) { if (int rc = ensure_ga()) return rc; }
    const bool has_sums = (mode != R_ACCEPT && mode != R_GRAD && mode != R_GRADT);
    if (chain()) {
        const double bytes = bytes_r(obj_->kind, mode, n, false);
        const bool big = bytes > big_bytes(mode == R_TRIAL || mode == R_UPG);
        const int grid = big ? GRID_BIG : grid_cg(n, 1);
        *grid_out = grid;
        if (int rc = prof_begin(kk)) return rc;
        if (int rc = launch_chain_kernel(mode, a_acc, beta, a, k, chain_npts(mode, npts), big, grid, make_tail(has_sums && !ctl && tail_fused(grid)))) return rc;
        return prof_end();
    }
    RParams P;
    P.x = xc_; P.u = uc_; P.gout = ga_.p; P.p0 = obj_->p0.p; P.n = n;
    P.xo = xc_; P.uo = uc_;
    P.a_acc = a_acc; P.beta = beta; P.s0 = obj_->s0; P.partials = ctx_->partials;
    P.ctl = ctl;
    P.x2 = xn_;
    for (int j = 0; j < MAXP; ++j) P.a[j] = (a && j < k) ? a[j] : ((a && k > 0) ? a[k - 1] : 0.0);
    const double bytes = bytes_r(obj_->kind, mode, n, obj_->uses_param());
    const bool big = bytes > big_bytes(mode == R_TRIAL || mode == R_UPG);
    const int grid = big ? GRID_BIG : grid_cg(n, npts);
    *grid_out = grid;
    P.tail = make_tail(has_sums && !ctl && tail_fused(grid));
    if (ctl && pipe_fused(grid)) {
        P.tail.partials2 = ctx_->partials2_f; P.tail.tickets = ctx_->tickets; P.tail.out = ctx_->out_dev;
        P.tail.strict = ctx_->tail_strict ? 1 : 0;
        P.tail.ctl = ctl_dev_; P.tail.ctl_rec = ctl_rec_; P.tail.ctl_seq = ctl_seq_;
        if (!ctx_->single()) {   // the finisher exchanges its block with the peers' GPUs itself (tail_exchange)
            P.tail.xw = ctx_->world(); P.tail.xme = ctx_->rank(); P.tail.xseq0 = epoch_ << 40;
            for (int r = 0; r < P.tail.xw && r < 8; ++r) P.tail.xmail[r] = ctx_->comm->dev_mailbox(r);
        }
    }
    if (P.tail.tickets) P.partials = ctx_->partials_f;
    const bool wr_x = (mode & R_ACCEPT) != 0, wr_u = (mode & (R_DIR | R_INIT | R_RESET)) != 0;
    const bool pp = big && !ctl && (wr_x || wr_u) && !(mode & R_PROJ) && pingpong_ready();
    if (pp && wr_x) P.xo = xalt_;
    if (pp && wr_u) P.uo = ualt_;
    if (mode == R_PROJ && !xn_) { set_error("internal: no second iterate buffer"); return CGO_ESTATE; }
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(kk)) return rc;
    int r = -2;
    switch (obj_->kind) {
    case CGO_OBJ_QUAD_DIAG: r = big ? launch_cg<ObjQuadDiag, true>(mode, npts, P, grid, st) : launch_cg<ObjQuadDiag, false>(mode, npts, P, grid, st); break;
    case CGO_OBJ_ROSENBROCK_PAIRED: r = big ? launch_cg<ObjRosenPaired, true>(mode, npts, P, grid, st) : launch_cg<ObjRosenPaired, false>(mode, npts, P, grid, st); break;
    case CGO_OBJ_BOOTH: r = big ? launch_cg<ObjBooth, true>(mode, npts, P, grid, st) : launch_cg<ObjBooth, false>(mode, npts, P, grid, st); break;
    case CGO_OBJ_USER:
        if (!obj_->rtc) { set_error("user objective has no compiled module"); return CGO_EINVAL; }
        if (int rc = launch_module(obj_->rtc->cg(mode, npts, big), &P, grid, st)) return rc;
        r = 0;
        break;
    default: break;
    }
    if (r) { set_error("internal: CG kernel mode not instantiated"); return CGO_EINVAL; }
    HIPCHK(hipGetLastError());
    if (pp && wr_x) std::swap(xc_, xalt_);
    if (pp && wr_u) std::swap(uc_, ualt_);
    return prof_end();
}

// ---- chained Rosenbrock: the stencil launches (cgo_kernels_chain.hip.hpp) ----------------------------------------
template <bool BIG>
static int launch_chain(int mode, int npts, const ChainParams &P, int grid, hipStream_t st) {
    switch (mode) {
    case R_INIT: k_chain<R_INIT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_TRIAL:
        if (npts == 3) k_chain<R_TRIAL, 3, BIG><<<grid, BLOCK, 0, st>>>(P);
        else k_chain<R_TRIAL, 1, BIG><<<grid, BLOCK, 0, st>>>(P);
        break;
    case R_ACCEPT | R_DIR | R_TRIAL:
        if (npts == 3) k_chain<R_ACCEPT | R_DIR | R_TRIAL, 3, BIG><<<grid, BLOCK, 0, st>>>(P);
        else k_chain<R_ACCEPT | R_DIR | R_TRIAL, 1, BIG><<<grid, BLOCK, 0, st>>>(P);
        break;
    case R_ACCEPT | R_DIR: k_chain<R_ACCEPT | R_DIR, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_ACCEPT: k_chain<R_ACCEPT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_RESET: k_chain<R_RESET, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_UPG: k_chain<R_UPG, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_GRAD: k_chain<R_GRAD, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_GRADT: k_chain<R_GRADT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_EDGES: k_chain<R_EDGES, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    default: return -1;
    }
    return 0;
}

int HipBackend::launch_chain_kernel(int mode, double a_acc, double beta, const double *a, int k, int npts, bool big, int grid, const Tail &tail) {
    ChainParams P;
    P.tail = tail;
    for (int j = 0; j < 3; ++j) P.a[j] = (a && j < k) ? a[j] : ((a && k > 0) ? a[k - 1] : 0.0);
    P.x = xc_; P.u = uc_; P.xo = xc_; P.uo = uc_; P.gout = ga_.p;
    P.odd = (int)(obj_->n_local & 1);
    P.n = obj_->n_local + P.odd; P.a_acc = a_acc; P.beta = beta; P.partials = tail.tickets ? ctx_->partials_f : ctx_->partials;
    for (int j = 0; j < 2; ++j) { P.hxl[j] = halo_xl_[j]; P.hul[j] = halo_ul_[j]; P.hxr[j] = halo_xr_[j]; P.hur[j] = halo_ur_[j]; }
    // the global vector ends where this rank's shard touches its ends
    P.has_left = obj_->offset > 0 ? 1 : 0;
    P.has_right = obj_->offset + obj_->n_local < obj_->n_global ? 1 : 0;
    const bool wr_x = (mode & R_ACCEPT) != 0, wr_u = (mode & (R_DIR | R_INIT | R_RESET)) != 0;
    if (wr_x) P.xo = xalt_;
    if (wr_u) P.uo = ualt_;
    const int r = big ? launch_chain<true>(mode, npts, P, grid, ctx_->stream) : launch_chain<false>(mode, npts, P, grid, ctx_->stream);
    if (r) { set_error("internal: chain kernel mode not instantiated"); return CGO_EINVAL; }
    HIPCHK(hipGetLastError());
    if (wr_x) std::swap(xc_, xalt_);
    if (wr_u) std::swap(uc_, ualt_);
    return CGO_OK;
}

// The instantiation a launch of kind `kk` uses under the current policy, as rocprofv3 prints it minus namespaces.
std::string HipBackend::kernel_symbol(int kk) const {
    const char *on = obj_->kind == CGO_OBJ_QUAD_DIAG ? "ObjQuadDiag" : obj_->kind == CGO_OBJ_ROSENBROCK_PAIRED ? "ObjRosenPaired"
                     : obj_->kind == CGO_OBJ_BOOTH ? "ObjBooth" : obj_->kind == CGO_OBJ_USER ? "UserObjective" : "";
    const int64_t n = obj_->n_local;
    const bool hp = obj_->uses_param();
    char buf[160];
    if (rmode_) {
        int mode = -1, npts = 1;
        switch (kk) {
        case KK_INIT: mode = R_INIT; break;
        case KK_TRIAL: mode = R_TRIAL; npts = npts_for(std::min(max_points(), 3)); break;
        case KK_ACCEPT_DIR_TRIAL: mode = R_ACCEPT | R_DIR | R_TRIAL; npts = npts_for(max_points()); break;
        case KK_ACCEPT_DIR: mode = R_ACCEPT | R_DIR; break;
        case KK_ACCEPT_ONLY: mode = R_ACCEPT; break;
        case KK_RESET_DIR: mode = R_RESET; break;
        case KK_UPG_NORM: mode = R_UPG; break;
        case KK_DIR_TRIAL: mode = R_DIR | R_TRIAL; npts = npts_for(max_points()); break;
        case KK_SYS_PROJECT: mode = R_PROJ; break;
        default: return "";
        }
        const bool big = bytes_r(obj_->kind, mode, n, hp) > big_bytes(mode == R_TRIAL || mode == R_UPG);
        if (chain()) snprintf(buf, sizeof buf, "k_chain<%d, %d, %s>", mode, chain_npts(mode, npts), big ? "true" : "false");
        else snprintf(buf, sizeof buf, "k_cg<%s, %d, %d, %s>", on, mode, npts, big ? "true" : "false");
        return buf;
    }
    if (obj_->two_phase()) {
        if (kk == KK_LSE_STATS) return "k_lse_stats";
        if (kk == KK_LSE_GRAD) return "k_lse_grad";
        const bool big_ring = 8.0 * (double)n * (3.0 + 2.0 * std::max(qn_m_ - 1, 0)) > big_bytes();
        if (kk == KK_LBFGS_FINAL && qn_m_ > 0) {   // the L-BFGS passes of the log-sum-exp objective (a full ring assumed for the policy bit)
            if (spec_on_ && qn_m_ - 1 <= SPEC_MAXC) { snprintf(buf, sizeof buf, "k_lbfgs_combine_spec<ObjLse, %s, %s>", big_ring ? "true" : "false", spec_fuse_push_ ? "true" : "false"); return buf; }
            if (gram_on_) { snprintf(buf, sizeof buf, "k_lbfgs_combine_lse<%s>", big_ring ? "true" : "false"); return buf; }
            return "k_lbfgs_loop";
        }
        if (kk == KK_LBFGS_PUSH && qn_m_ > 0) {
            if (spec_on_ && qn_m_ - 1 <= SPEC_MAXC && !spec_fuse_push_) { snprintf(buf, sizeof buf, "k_lbfgs_push_lite<ObjLse, %s>", 8.0 * (double)n * 7.0 > big_bytes() ? "true" : "false"); return buf; }
            if (gram_on_) { snprintf(buf, sizeof buf, fuse_grad_ && x2_.p ? "k_lbfgs_push_gram_lse<%s>" : "k_lbfgs_push_gram<%s>", big_ring ? "true" : "false"); return buf; }
            return "k_lbfgs_push";
        }
        return "";
    }
    int mode = -1;
    switch (kk) {
    case KK_INIT: mode = M_INIT; break;
    case KK_TRIAL: mode = need_beta_ ? (M_TRIAL | M_BETA) : M_TRIAL; break;
    case KK_ACCEPT_DIR_TRIAL: mode = M_ACCEPT | M_DIR | M_TRIAL | M_BETA; break;
    case KK_ACCEPT_DIR: mode = M_ACCEPT | M_DIR; break;
    case KK_ACCEPT_ONLY: mode = M_ACCEPT; break;
    case KK_RESET_DIR: mode = M_RESET; break;
    case KK_UPG_NORM: mode = M_UPG; break;
    case KK_LBFGS_PUSH: return gram_on_ ? "k_lbfgs_push_gram" : "k_lbfgs_push";
    case KK_LBFGS_LOOP: return "k_lbfgs_loop";
    case KK_LBFGS_FINAL:
        if (spec_on_ && qn_m_ > 0 && qn_m_ - 1 <= SPEC_MAXC) {   // the one-pass form (a full ring assumed for the policy bit)
            snprintf(buf, sizeof buf, "k_lbfgs_combine_spec<%s, %s, %s>", on, 8.0 * (double)n * (3.0 + (hp ? 1.0 : 0.0) + 2.0 * (qn_m_ - 1)) > big_bytes() ? "true" : "false",
                     spec_fuse_push_ ? "true" : "false");
            return buf;
        }
        return gram_on_ ? "k_lbfgs_combine" : "k_lbfgs_loop";
    default: return "";
    }
    const bool objective_mode = (mode & (M_TRIAL | M_INIT)) != 0;
    snprintf(buf, sizeof buf, "k_fused<%s, %d, %s>", objective_mode ? on : "ObjQuadDiag", mode, is_big(obj_->kind, mode, n, hp, pol_.hbm_stream_bytes) ? "true" : "false");
    return buf;
}

// ---- on-device controller (cgo_ctl.hpp) ------------------------------------------------------
// Device block: the controller's config and state, and the argument block the armed launches read.
// `round` numbers the rounds of a solve on the DEVICE: the reduce/controller kernel derives its record slot and its
// sequence word from it, so that a round's kernels carry no per-round host argument at all and whole batches of
// rounds replay from one instantiated hipGraph (pipe_launch_graph).
// (struct CtlDev: cgo_kernels_cg.hip.hpp — the armed launches' own finisher reads and writes it too)

__global__ void k_ctl_init(CtlDev *d, const CtlConfig cfg, const CtlState st, unsigned long long round) {
    d->cfg = cfg;
    d->st = st;
    d->round = round;
    CtlArgs a;
    a.a_acc = st.a_acc; a.beta = st.beta; a.go = st.go;
    for (int j = 0; j < CTL_MAXP; ++j) a.a[j] = st.a[j];
    d->args = a;
}

// Final reduction stage of a controller-armed launch + the controller itself: rows → sums →
// ctl_step() → arguments of the next launch (device memory) and the round's record (pinned host
// memory, released with a sequence word the host polls).
// One lane running scalar code is the slow part of this kernel (a dependent global load costs ≈ 1–2 µs, a
// PCIe store ≈ 0.2 µs): the device block is staged into LDS and the results are written back — state and
// arguments to HBM, the 30-word record to pinned host memory — by as many lanes as there are words.


template <int N, int THREADS>
__global__ __launch_bounds__(THREADS) void k_finalize_ctl(const double *partials, int rows, double *out, CtlDev *d,
                                                          CtlRecord *rec_ring, unsigned long long *seq_ring) {
    constexpr int G = BLOCK / N;   // the k_cg family's summation order (finalize_rows canon, finish_tail)
    constexpr int WD = sizeof(CtlDev) / 8, WR = sizeof(CtlRecord) / 8;
    __shared__ double sm[G][N];
    __shared__ double fin[CTL_NSUMS];
    __shared__ CtlDev sd;
    __shared__ CtlRecord sr;
    const int tid = threadIdx.x;
    if (tid < WD) ((unsigned long long *)&sd)[tid] = ((const unsigned long long *)d)[tid];
    if (tid < CTL_NSUMS) fin[tid] = 0.0;
    __syncthreads();
    const bool go = sd.st.go != 0;
    const unsigned long long round = sd.round, seq = round + 1;
    CtlRecord *rec_host = rec_ring + (round % PIPE_RING);
    unsigned long long *seq_host = seq_ring + (round % PIPE_RING);
    if (go) {  // same summation order as k_finalize_t: the record must hold what a host-driven launch would
        if (tid < G * N) {
            double t = 0.0;
            const long long total = (long long)rows * N;
            for (long long i = tid; i < total; i += G * N) t += partials[i];
            sm[tid / N][tid % N] = t;
        }
        __syncthreads();
        if (tid < N) {
            double v = 0.0;
#pragma unroll
            for (int g = 0; g < G; ++g) v += sm[g][tid];
            out[tid] = v;
            fin[tid] = v;
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (go) {
            ctl_step(sd.cfg, sd.st, fin, sr);
            CtlArgs a;
            a.a_acc = sd.st.a_acc; a.beta = sd.st.beta; a.go = sd.st.go;
            for (int j = 0; j < CTL_MAXP; ++j) a.a[j] = sd.st.a[j];
            sd.args = a;
        } else {
            for (int i = 0; i < CTL_NSUMS; ++i) sr.sums[i] = 0.0;
            sr.a_acc = 0.0; sr.beta = 0.0;
            for (int j = 0; j < CTL_MAXP; ++j) sr.a[j] = 0.0;
            sr.npts = -1; sr.accepted = 0; sr.xwait = 0;
        }
        sd.round = round + 1;
    }
    __syncthreads();
    if (go && tid < WD) ((unsigned long long *)d)[tid] = ((const unsigned long long *)&sd)[tid];
    if (!go && tid == 0) d->round = round + 1;
    if (tid < WR) {
        ((unsigned long long *)rec_host)[tid] = ((const unsigned long long *)&sr)[tid];
        __threadfence_system();
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Armed rounds: one rank — or several whose GPUs exchange their blocks themselves (device mailboxes, cgo_comm.hip), which only
// the single-launch form of a round does (tail_ctl): built-in objective, grid-stride launch with a fused tail.
int HipBackend::ctl_depth() const {
    if (!(rmode_ && ctx_->host_publish && !obj_->two_phase())) return 0;
    if (ctx_->single()) return ctl_depth_;
    if (!ctx_->dev_exchange() || ctx_->force_gather) return 0;
    const bool big = bytes_r(obj_->kind, R_ACCEPT | R_DIR | R_TRIAL, obj_->n_local, obj_->uses_param()) > big_bytes(false);
    return (!big && pipe_fused(grid_cg(obj_->n_local, policy_points()))) ? ctl_depth_ : 0;
}

int HipBackend::pipe_alloc() {
    if (ctl_dev_) return CGO_OK;
    HIPCHK(hipMalloc(&ctl_dev_, sizeof(CtlDev)));
    HIPCHK(hipHostMalloc((void **)&ctl_rec_, sizeof(CtlRecord) * PIPE_RING, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&ctl_seq_, sizeof(unsigned long long) * PIPE_RING, hipHostMallocDefault));
    std::memset(ctl_rec_, 0, sizeof(CtlRecord) * PIPE_RING);
    std::memset(ctl_seq_, 0, sizeof(unsigned long long) * PIPE_RING);
    pipe_prof_.assign(PIPE_RING, {-1, 0u});
    return CGO_OK;
}

// At solver creation: the controller's blocks, and ONE armed round with the controller stopped — a no-op that files an idle
// record — so that the first launches of k_ctl_init and of the armed kernel (≈ 60 µs each of code-object set-up) do not
// fall into the first armed iteration (BASELINE config 1 runs 25 iterations in all: 14.4k vs 16.5k it/s).
int HipBackend::prepare_controller() {
    if (ctl_depth() <= 0) return CGO_OK;
    if (int rc = pipe_alloc()) return rc;
    if (obj_->uses_param() && !obj_->p0_set) return CGO_OK;   // nothing to launch on yet
    HIPCHK(hipSetDevice(ctx_->device));
    CtlConfig cc{};
    CtlState st{};
    st.go = 0;
    pipe_npts_ = max_points();
    k_ctl_init<<<1, 1, 0, ctx_->stream>>>((CtlDev *)ctl_dev_, cc, st, pipe_enq_);
    HIPCHK(hipGetLastError());
    if (int rc = pipe_enqueue_round()) return rc;
    return pipe_drain();
}

// the kernels of one controller-armed round: k_cg reading its scalars from the device block, then reduce + controller.
// No argument depends on the round (record slot and sequence number come from CtlDev::round), so the same launches can
// be captured into a hipGraph.
int HipBackend::pipe_round_kernels() {
    int grid = 0;
    const int npts = pipe_npts_, ns = rows_for(npts);
    CtlDev *d = (CtlDev *)ctl_dev_;
    if (int rc = launch_r_kernel(KK_ACCEPT_DIR_TRIAL, R_ACCEPT | R_DIR | R_TRIAL, 0.0, 0.0, nullptr, 0, npts, &d->args, &grid)) return rc;
    pipe_checked_ = pipe_fused(grid) && !ctx_->tail_strict;
    if (pipe_fused(grid)) return CGO_OK;   // the launch's own finisher reduced, ran the controller and published the record
    hipStream_t st = ctx_->stream;
    const double *src = ctx_->partials;
    int nrows = grid;
    if (grid > TAIL_GROUP) {
        const int nb = (grid + TAIL_GROUP - 1) / TAIL_GROUP;
        if (ns == NR) k_finalize_t<NR, BLOCK><<<nb, BLOCK, 0, st>>>(ctx_->partials, TAIL_GROUP, grid, ctx_->partials2, nullptr, nullptr, 0);
        else if (ns == NR5) k_finalize_t<NR5, BLOCK><<<nb, BLOCK, 0, st>>>(ctx_->partials, TAIL_GROUP, grid, ctx_->partials2, nullptr, nullptr, 0);
        else if (ns == NR7) k_finalize_t<NR7, BLOCK><<<nb, BLOCK, 0, st>>>(ctx_->partials, TAIL_GROUP, grid, ctx_->partials2, nullptr, nullptr, 0);
        else k_finalize_t<NS, BLOCK><<<nb, BLOCK, 0, st>>>(ctx_->partials, TAIL_GROUP, grid, ctx_->partials2, nullptr, nullptr, 0);
        HIPCHK(hipGetLastError());
        src = ctx_->partials2;
        nrows = nb;
    }
    CtlRecord *rec = (CtlRecord *)ctl_rec_;
    if (ns == NR) k_finalize_ctl<NR, 768><<<1, 768, 0, st>>>(src, nrows, ctx_->out_dev, d, rec, ctl_seq_);
    else if (ns == NR5) k_finalize_ctl<NR5, 768><<<1, 768, 0, st>>>(src, nrows, ctx_->out_dev, d, rec, ctl_seq_);
    else if (ns == NR7) k_finalize_ctl<NR7, 768><<<1, 768, 0, st>>>(src, nrows, ctx_->out_dev, d, rec, ctl_seq_);
    else k_finalize_ctl<NS, BLOCK><<<1, BLOCK, 0, st>>>(src, nrows, ctx_->out_dev, d, rec, ctl_seq_);
    HIPCHK(hipGetLastError());
    return CGO_OK;
}

// one round, launched kernel by kernel (with a HIP-event sample when the profiler picks it)
int HipBackend::pipe_enqueue_round() {
    if (int rc = pipe_round_kernels()) return rc;
    const int idx = (int)(pipe_enq_ % PIPE_RING);
    pipe_prof_[idx] = {prof_cur_ ? ring_used_ - 1 : -1, prof_gen_};
    prof_cur_ = false;
    pipe_enq_++;
    return CGO_OK;
}

// `rounds` rounds as ONE hipGraphLaunch: the per-launch host cost (≈ 3.5 µs per kernel, two or three kernels per round)
// is what kept the device waiting for the host at small n although the controller needs no host decision
// (DESIGN.md §2.7).  Instantiated once per (rounds, row width, buffers) and replayed.
int HipBackend::pipe_launch_graph(int rounds) {
    HIPCHK(hipSetDevice(ctx_->device));
    PipeGraph *g = nullptr;
    for (auto &c : graphs_)
        if (c.rounds == rounds && c.npts == pipe_npts_ && c.x == xc_ && c.u == uc_ && c.p0 == obj_->p0.p && c.n == obj_->n_local) { g = &c; break; }
    if (!g) {
        hipStream_t st = ctx_->stream;
        hipGraph_t graph = nullptr;
        capturing_ = true;
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        int rc = CGO_OK;
        if (e == hipSuccess) {
            for (int r = 0; r < rounds && rc == CGO_OK; ++r) rc = pipe_round_kernels();
            hipError_t e2 = hipStreamEndCapture(st, &graph);
            if (e2 != hipSuccess) e = e2;
        }
        capturing_ = false;
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || !graph) { set_error(std::string("hipGraph capture of controller rounds failed: ") + hipGetErrorString(e)); (void)hipGetLastError(); return CGO_EHIP; }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { set_error(std::string("hipGraphInstantiate failed: ") + hipGetErrorString(e)); return CGO_EHIP; }
        graphs_.push_back(PipeGraph{exec, rounds, pipe_npts_, xc_, uc_, obj_->p0.p, obj_->n_local});
        g = &graphs_.back();
    }
    HIPCHK(hipGraphLaunch((hipGraphExec_t)g->exec, ctx_->stream));
    for (int r = 0; r < rounds; ++r) {
        pipe_prof_[(int)(pipe_enq_ % PIPE_RING)] = {-1, prof_gen_};
        pipe_enq_++;
    }
    graph_rounds_ += rounds;
    return CGO_OK;
}

// enqueue `count` more rounds: graphs of 8 / 4 / 2 rounds where possible.  With the profiler on, every 4th batch goes
// kernel by kernel so that the HIP-event samples of the armed launches keep coming.
int HipBackend::pipe_enqueue(int64_t count) {
    const bool eager = !graph_on_ || (prof_on_ && ((pipe_batches_++ & 3) == 0));
    while (count > 0) {
        int c = 1;
        if (!eager) { c = 8; while (c > count) c >>= 1; }
        if (c == 1) { if (int rc = pipe_enqueue_round()) return rc; }
        else if (int rc = pipe_launch_graph(c)) return rc;
        count -= c;
    }
    return CGO_OK;
}

// wait for the record of global round `id` (0-based)
int HipBackend::pipe_wait(unsigned long long id, CtlRecord &rec) {
    const int idx = (int)(id % PIPE_RING);
    if (pipe_checked_) {   // fused rounds: the record validates itself (tail_publish_record)
        static_assert(sizeof(CtlRecord) % 8 == 0, "record = 8-byte words");
        constexpr int WR = (int)(sizeof(CtlRecord) / 8);
        double words[WR];
        if (int rc = wait_checked(ctx_, ctl_seq_ + idx, id + 1, reinterpret_cast<const double *>(ctl_rec_) + (size_t)idx * WR, WR, words)) return rc;
        std::memcpy(&rec, words, sizeof(CtlRecord));
        if (rec.npts >= 0 && !ctx_->single()) {   // the round exchanged its block between the GPUs: its cost, for cgo_ctx_exchange_stats
            ctx_->xch_count++; ctx_->xch_dev_ms += (double)rec.xwait * 1e-5; ctx_->xch_dev_n++;
        }
        return CGO_OK;
    }
    if (int rc = wait_word(ctx_, ctl_seq_ + idx, id + 1)) return rc;
    rec = ((CtlRecord *)ctl_rec_)[idx];
    return CGO_OK;
}

// Before any launch that is not controller-armed: every round still in flight must be a no-op
// (the controller stops exactly where the host-side state machine leaves the fast path).
int HipBackend::pipe_drain() {
    while (pipe_done_ < pipe_enq_) {
        CtlRecord rec;
        if (int rc = pipe_wait(pipe_done_, rec)) return rc;
        pipe_done_++;
        if (rec.npts >= 0) {
            set_error("internal: the on-device controller ran a launch the host state machine did not ask for");
            return CGO_ESTATE;
        }
    }
    return CGO_OK;
}

int HipBackend::accept_dir_trial_ctl(const CtlConfig &cc, const CtlState &s0, int64_t rounds, Scal *out) {
    if (ctl_depth() <= 0) return accept_dir_trial(s0.a_acc, s0.beta, s0.a, s0.npts, out);
    if (int rc = pipe_alloc()) return rc;
    // how far to run ahead: one more round per first trial accepted in a row (host-observed)
    const int64_t ahead = std::min<int64_t>(std::min<int64_t>(ctl_depth_, pipe_streak_), rounds - 1);
    if (pipe_done_ == pipe_enq_) {  // idle: arm a new batch from the host's state
        if (ahead <= 0) { pipe_streak_++; return accept_dir_trial_keep_streak(s0, out); }
        HIPCHK(hipSetDevice(ctx_->device));
        pipe_npts_ = cc.maxp;
        k_ctl_init<<<1, 1, 0, ctx_->stream>>>((CtlDev *)ctl_dev_, cc, s0, pipe_enq_);
        HIPCHK(hipGetLastError());
        pipe_stopped_ = false;
        if (int rc = pipe_enqueue(1 + ahead)) return rc;
    }
    CtlRecord rec;
    const unsigned long long id = pipe_done_;
    if (int rc = pipe_wait(id, rec)) return rc;
    pipe_done_++;
    if (rec.npts < 0) {  // the controller had stopped before this round: the host drives it
        if (int rc = pipe_drain()) return rc;
        pipe_streak_++;
        return accept_dir_trial_keep_streak(s0, out);
    }
    if (std::memcmp(&rec.a_acc, &s0.a_acc, 8) || std::memcmp(&rec.beta, &s0.beta, 8) || rec.npts != s0.npts ||
        std::memcmp(rec.a, s0.a, 8 * (size_t)s0.npts)) {
        set_error("internal: the on-device controller and the host state machine disagree on a launch");
        return CGO_ESTATE;
    }
    const int np = pipe_npts_;
    for (int j = 0; j < s0.npts; ++j) {
        const double *q = rec.sums + RS_PER_POINT * j;
        out[j].f = q[RS_F]; out[j].gtu = q[RS_GTU]; out[j].gtgt = q[RS_GTGT]; out[j].gtg = q[RS_GTG];
        out[j].yy = q[RS_YY]; out[j].uy = q[RS_UY]; out[j].ygt = q[RS_YGT];
    }
    out[0].gu = rec.sums[RS_PER_POINT * np]; out[0].uu = rec.sums[RS_PER_POINT * np + 1];
    total_launches_++;
    pipe_served_++;
    pipe_streak_++;
    const auto &pp = pipe_prof_[(int)(id % PIPE_RING)];
    if (prof_on_) {
        prof_cnt_[KK_ACCEPT_DIR_TRIAL]++;
        prof_bytes_[KK_ACCEPT_DIR_TRIAL] = bytes_r(obj_->kind, R_ACCEPT | R_DIR | R_TRIAL, obj_->n_local, obj_->uses_param());
        if (pp.first >= 0 && pp.second == prof_gen_ && pp.first < ring_used_) {
            ring_[pp.first].kk = KK_ACCEPT_DIR_TRIAL;
            ring_[pp.first].bytes = prof_bytes_[KK_ACCEPT_DIR_TRIAL];
        }
    }
    if (!rec.accepted) pipe_stopped_ = true;
    if (!pipe_stopped_) {  // keep the device `ahead` rounds in front of the host
        // top the run-ahead up in batches (half the depth at a time) so that graph replays stay worth their launch
        const int64_t want = std::min<int64_t>(std::min<int64_t>(ctl_depth_, pipe_streak_), rounds - 1);
        const int64_t have = (int64_t)(pipe_enq_ - pipe_done_);
        if (have < want && (want - have >= (want + 1) / 2 || have == 0))
            if (int rc = pipe_enqueue(want - have)) return rc;
    }
    return CGO_OK;
}

// host-driven accept+dir+trial that does not reset the first-trial streak counter
int HipBackend::accept_dir_trial_keep_streak(const CtlState &s0, Scal *out) {
    const int64_t keep = pipe_streak_;
    const int rc = accept_dir_trial(s0.a_acc, s0.beta, s0.a, s0.npts, out);
    pipe_streak_ = keep;
    return rc;
}

static void unpack_r(const double *s, int k, Scal *out, bool dir) {
    const int npts = npts_for(k);
    for (int j = 0; j < k; ++j) {
        const double *q = s + RS_PER_POINT * j;
        out[j].f = q[RS_F]; out[j].gtu = q[RS_GTU]; out[j].gtgt = q[RS_GTGT]; out[j].gtg = q[RS_GTG];
        out[j].yy = q[RS_YY]; out[j].uy = q[RS_UY]; out[j].ygt = q[RS_YGT];
    }
    if (dir) { out[0].gu = s[RS_PER_POINT * npts]; out[0].uu = s[RS_PER_POINT * npts + 1]; }
}

// ---- host-closure objective (cgo_objective_create_callback) ----------------------------------
// evalϕdϕ! (cg_utils.jl:4-23) around the user's f = fdf!(g, x): the trial point is formed on the device (unfused,
// bit-identical to the reference's loop) and stored straight into pinned host memory, the closure runs on the host,
// g⁺ returns to the device, and ONE launch reduces every sum the line search and getβ need from (g⁺, g, u); the
// closure's f rides in that launch's S_F slot so that it crosses ranks with the rest.  init: x itself, then u = −g.
int HipBackend::host_trial(double a, bool init, Scal &out) {
    HIPCHK(hipSetDevice(ctx_->device));
    const int64_t n = obj_->n_local;
    if (!obj_->host_fn || !obj_->host_x || !obj_->host_g) { set_error("host objective has no callback"); return CGO_ESTATE; }
    hipStream_t st = ctx_->stream;
    int grid = (int)std::min<int64_t>((n + BLOCK - 1) / BLOCK, GRID_SMALL);
    if (grid < 1) grid = 1;
    k_trial_point<<<grid, BLOCK, 0, st>>>(xc_, init ? nullptr : u_.p, a, obj_->host_x, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    const double f_local = obj_->host_fn(obj_->host_user, obj_->host_g, obj_->host_x, n);
    HIPCHK(hipMemcpyAsync(gt_, obj_->host_g, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    total_launches_++;
    if (init) {   // g and u hold nothing yet: zero them so that the unused sums of this launch stay finite
        HIPCHK(hipMemsetAsync(g_, 0, sizeof(double) * (size_t)n, st));
        HIPCHK(hipMemsetAsync(u_.p, 0, sizeof(double) * (size_t)n, st));
    }
    double s[NS];
    if (int rc = launch(KK_TRIAL, M_BETAONLY, 0, 0, f_local, true, s)) return rc;
    if (init) {
        out = Scal();
        out.f = s[S_F]; out.gtgt = s[S_GTGT];
        std::swap(g_, gt_);   // the gradient just received becomes the current one
        Scal d;
        return reset_dir(d);  // info.u = −df_x  (cg_flavours.jl:29)
    }
    out.f = s[S_F]; out.gtu = s[S_GTU]; out.gtgt = s[S_GTGT]; out.gtg = s[S_GTG];
    out.yy = s[S_YY]; out.uy = s[S_UY]; out.ygt = s[S_YGT];
    return CGO_OK;
}

// ---- two-phase objective (log-sum-exp) --------------------------------------------------
template <int MODE>
static int launch_lse_stats(const LseParams &P, bool big, bool ref, int grid, hipStream_t st) {
    if (ref) {
        if (big) k_lse_stats<MODE, true, true><<<grid, BLOCK, 0, st>>>(P);
        else k_lse_stats<MODE, false, true><<<grid, BLOCK, 0, st>>>(P);
    } else {
        if (big) k_lse_stats<MODE, true, false><<<grid, BLOCK, 0, st>>>(P);
        else k_lse_stats<MODE, false, false><<<grid, BLOCK, 0, st>>>(P);
    }
    return 0;
}

int HipBackend::lse_stats(int mode, double a_acc, double beta, double a_trial, Scal &out, bool dir) {
    if (int rc = flush_lite()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    const int64_t n = obj_->n_local;
    LseParams P;
    P.x = xc_; P.u = u_.p; P.g = g_; P.gt = gt_; P.n = n;
    P.a_acc = a_acc; P.beta = beta; P.a_trial = a_trial; P.lambda = obj_->s0; P.M = 0; P.S = 1;
    P.partials = ctx_->partials;
    // Fixed-reference form (k_lse_stats<…, REF>): the reference is lse of the last point evaluated on this line (for the fused
    // accept + direction + trial launch: of the iterate being accepted).  Not for the very first evaluation (no reference yet).
    const bool ref = pol_.lse_fixed_reference != 0 && mode != LM_NOU && lse_have_;
    const double Mr = ref ? lse_M_ + std::log(lse_S_) : 0.0;
    if (ref) P.M = Mr;
    const double nvec = (mode == LM_NOU) ? 1.0 : (mode == 0 ? 2.0 : 5.0);
    const double bytes = 8.0 * (double)n * nvec;
    const bool big = bytes > big_bytes(mode == 0 || mode == LM_NOU);
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LSE_STATS)) return rc;
    if (mode == 0) launch_lse_stats<0>(P, big, ref, grid, st);
    else if (mode == LM_NOU) launch_lse_stats<LM_NOU>(P, big, ref, grid, st);
    else launch_lse_stats<LM_ACCEPT | LM_DIR>(P, big, ref, grid, st);
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_launch(ctx_, grid, !ref)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s, ref ? MERGE_SUM : MERGE_LSE)) return rc;
    if (prof_on_) prof_commit(KK_LSE_STATS, bytes);
    if (dir) { out.gu = s[S_GU]; out.uu = s[S_UU]; }
    if (ref) {
        const double Sp = s[L_S];
        if (!(Sp >= 1e-280 && Sp <= 1e280) || !std::isfinite(s[L_T])) {   // the trial is far from the reference: take it again from its own maximum
            lse_have_ = false;                                               // (x, u are already updated if this was a fused launch: a plain trial now)
            Scal t;
            if (int rc = lse_stats(0, 0, 0, a_trial, t, false)) return rc;
            out.f = t.f; out.gtu = t.gtu;
            return CGO_OK;
        }
        lse_a_ = a_trial; lse_M_ = Mr; lse_S_ = Sp;   // (M_r, S') describe xp as well as its own (max, Σ) would
        out.f = (Mr + std::log(Sp)) + 0.5 * obj_->s0 * s[L_Q];
        out.gtu = s[L_T] / Sp + obj_->s0 * s[L_R];
        return CGO_OK;
    }
    lse_a_ = a_trial; lse_M_ = s[L_M]; lse_S_ = s[L_S];
    lse_have_ = std::isfinite(lse_M_) && lse_S_ > 0.0 && std::isfinite(lse_S_);
    out.f = (s[L_M] + std::log(s[L_S])) + 0.5 * obj_->s0 * s[L_Q];  // ϕ = lse + ½λ‖xp‖²
    out.gtu = s[L_T] / s[L_S] + obj_->s0 * s[L_R];                   // dϕ = softmax·u + λ xp·u
    return CGO_OK;
}

int HipBackend::lse_grad(bool init, double a, Scal &out) {
    if (int rc = flush_lite()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    const int64_t n = obj_->n_local;
    LseParams P;
    P.x = xc_; P.u = u_.p; P.g = g_; P.gt = gt_; P.n = n;
    P.a_acc = 0; P.beta = 0; P.a_trial = a; P.lambda = obj_->s0; P.M = lse_M_; P.S = lse_S_;
    P.partials = ctx_->partials;
    const bool beta = need_beta_ && !init;
    const double bytes = 8.0 * (double)n * (init ? 3.0 : (beta ? 4.0 : 3.0));
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LSE_GRAD)) return rc;
    if (init) { if (big) k_lse_grad<false, true, true><<<grid, BLOCK, 0, st>>>(P); else k_lse_grad<false, true, false><<<grid, BLOCK, 0, st>>>(P); }
    else if (beta) { if (big) k_lse_grad<true, false, true><<<grid, BLOCK, 0, st>>>(P); else k_lse_grad<true, false, false><<<grid, BLOCK, 0, st>>>(P); }
    else { if (big) k_lse_grad<false, false, true><<<grid, BLOCK, 0, st>>>(P); else k_lse_grad<false, false, false><<<grid, BLOCK, 0, st>>>(P); }
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_launch(ctx_, grid, false)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s)) return rc;
    if (prof_on_) prof_commit(KK_LSE_GRAD, bytes);
    out.gtgt = s[S_GTGT]; out.gtg = s[S_GTG]; out.yy = s[S_YY]; out.uy = s[S_UY]; out.ygt = s[S_YGT];
    return CGO_OK;
}

int HipBackend::materialize(Scal &out) {
    if (obj_->two_phase()) return lse_grad(false, lse_a_, out);
    if (!spec_unmat_) return CGO_OK;   // element-wise objectives: every trial launch writes its g⁺ …
    const double a = spec_a_;          // … except the trial a direction pass speculated on, when its sums could not be used for the push
    return trial(&a, 1, &out);
}

// ---- L-BFGS ring in HBM ------------------------------------------------------------------
int HipBackend::lbfgs_alloc(int m) {  // m = number of physical slots (history + 1)
    HIPCHK(hipSetDevice(ctx_->device));
    qn_m_ = m;
    gram_on_ = pol_.lbfgs_form != 4;     // 4: chained two-loop launches
    if (m - 1 > GRAM_MAXC) gram_on_ = false;
    const size_t n = (size_t)obj_->n_local;
    if (int rc = qn_S_.alloc(n * (size_t)m)) return rc;
    if (int rc = qn_Y_.alloc(n * (size_t)m)) return rc;
    if (qn_alpha_dev_) (void)hipFree(qn_alpha_dev_);
    HIPCHK(hipMalloc((void **)&qn_alpha_dev_, sizeof(double) * 64));
    qn_sgt_slot_ = -1;
    push_pending_ = false; push_lite_pending_ = false; spec_valid_ = false;
    {   // lbfgs_form 0 / 1: one ring pass, state update riding in the next pass · 2: its own launch · 3: Gram form, two passes
      const bool capable = obj_->two_phase() || (!rmode_ && (obj_->kind == CGO_OBJ_QUAD_DIAG || obj_->kind == CGO_OBJ_ROSENBROCK_PAIRED ||
                                                             (obj_->kind == CGO_OBJ_USER && obj_->rtc && obj_->rtc->spec(false, false))));
      spec_on_ = pol_.lbfgs_form <= 2 && gram_on_ && capable; spec_fuse_push_ = pol_.lbfgs_form != 2; }
    spec_unmat_ = false;
    lite_deferred_ = false;
    // the second iterate buffer of the fused push (lbfgs_push_materializes); a rank of a sharded solve that cannot have it
    // fails here rather than falling out of step with its peers, a single rank just keeps the two-launch form
    fuse_grad_ = pol_.lbfgs_fuse_grad != 0;
    if (fuse_grad_ && gram_on_ && obj_->two_phase() && m - 1 <= GRAM_MAXC_LSE && !x2_.p) {
        const int rc = x2_.alloc(n);
        if (rc != CGO_OK && ctx_->world() > 1) return rc;
        if (rc != CGO_OK) (void)hipGetLastError();
    }
    return CGO_OK;
}

// ---- Gram ("vector-free") form of the L-BFGS update ------------------------------------------
// Log-sum-exp objective: the push forms g⁺ of the accepted trial itself (k_lbfgs_push_gram<…, true>) — no k_lse_grad launch.
// Needs a second iterate buffer (x advances out of place until lbfgs_push_commit) and a free row slot for Σ g⁺² (m ≤ 11).
// CGO_LBFGS_FUSE_GRAD=0 keeps materialize() + the plain push (A/B).
bool HipBackend::lbfgs_push_materializes(double a_x) {
    if (!fuse_grad_ || !gram_on_ || !obj_->two_phase() || qn_m_ - 1 > GRAM_MAXC_LSE) return false;
    if (std::memcmp(&a_x, &lse_a_, sizeof(double)) != 0) return false;   // the statistics at hand are those of another step
    return x2_.p != nullptr;   // (lbfgs_alloc: every rank has it or the solve did not start — the ranks' launch sequences must agree)
}

// direction_follows: the caller's next call is the direction of the following iteration — a speculated push then rides in that
// pass (k_lbfgs_combine_spec<…, PUSH>) instead of a launch of its own.  Whatever else touches x, g or the ring first
// (flush_lite at the head of every such entry point) runs the state update as its own launch.
int HipBackend::lbfgs_push_commit(bool direction_follows) {
    if (push_lite_pending_) {
        push_lite_pending_ = false;
        if (direction_follows && spec_fuse_push_) { lite_deferred_ = true; return CGO_OK; }
        return lbfgs_push_lite();
    }
    if (!push_pending_) return CGO_OK;
    push_pending_ = false;
    xc_ = push_xo_;
    std::swap(g_, gt_);  // g ← g⁺
    return CGO_OK;
}

int HipBackend::lbfgs_push_gram(double a_x, double a_s, int slot, const int *prev, int count, GramOut &out) {
    if (int rc = flush_lite()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    if (count > GRAM_MAXC) { set_error("internal: Gram form limited to 12 pairs"); return CGO_EINVAL; }
    const bool fused = lbfgs_push_materializes(a_x) && count <= GRAM_MAXC_LSE;
    const int64_t n = obj_->n_local;
    GramPushParams P;
    P.x = xc_; P.u = u_.p; P.g = g_; P.gt = gt_; P.S = qn_S_.p; P.Y = qn_Y_.p;
    P.n = n; P.a = a_x; P.a_s = a_s; P.slot = slot; P.count = count; P.partials = ctx_->partials;
    for (int j = 0; j < GRAM_MAXC; ++j) P.prev[j] = j < count ? prev[j] : 0;
    GramLseParams L{};
    if (fused) {
        push_xo_ = (xc_ == x_.p) ? x2_.p : x_.p;
        L.xo = push_xo_; L.gt_out = gt_; L.M = lse_M_; L.S = lse_S_; L.lambda = obj_->s0;
    }
    const double bytes = 8.0 * (double)n * (7.0 + 2.0 * count);   // fused: R x,u,g + ring, W x',s,y,g⁺ — the same count, g⁺ written instead of read
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LBFGS_PUSH)) return rc;
    if (fused) {
        if (big) k_lbfgs_push_gram_lse<true><<<grid, BLOCK, 0, st>>>(P, L);
        else k_lbfgs_push_gram_lse<false><<<grid, BLOCK, 0, st>>>(P, L);
    } else {
        if (big) k_lbfgs_push_gram<true><<<grid, BLOCK, 0, st>>>(P);
        else k_lbfgs_push_gram<false><<<grid, BLOCK, 0, st>>>(P);
    }
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_rows(ctx_, grid, NG)) return rc;
    double s[NG];
    if (int rc = fetch_sums(ctx_, s, MERGE_SUM, NG)) return rc;
    if (prof_on_) prof_commit(KK_LBFGS_PUSH, bytes);
    out.sy = s[0]; out.yy = s[1]; out.sgn = s[2]; out.ygn = s[3];
    for (int j = 0; j < count; ++j) {
        out.sjg[j] = s[4 + 5 * j]; out.yjg[j] = s[5 + 5 * j]; out.sjyn[j] = s[6 + 5 * j];
        out.yjsn[j] = s[7 + 5 * j]; out.yjyn[j] = s[8 + 5 * j];
    }
    qn_sgt_slot_ = -1;
    out.materialized = fused;
    push_counts_[fused ? 1 : 2]++;
    if (fused) {          // x, g stay the last good iterate until the caller has seen ‖g⁺‖ (optim.jl:107-121): lbfgs_push_commit
        out.gtgt = s[GRAM_GTGT];
        push_pending_ = true;
    } else {
        std::swap(g_, gt_);  // g ← g⁺
    }
    return CGO_OK;
}

int HipBackend::lbfgs_direction_gram(const int *slots, const double *cy, const double *cs, int count, double cg,
                                     Scal &out) {
    if (int rc = flush_lite()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    if (count > GRAM_MAXC) { set_error("internal: Gram form limited to 12 pairs"); return CGO_EINVAL; }
    const int64_t n = obj_->n_local;
    GramDirParams P;
    P.g = g_; P.u = u_.p; P.S = qn_S_.p; P.Y = qn_Y_.p; P.n = n; P.count = count; P.cg = cg;
    P.partials = ctx_->partials;
    for (int j = 0; j < GRAM_MAXC; ++j) {
        P.slots[j] = j < count ? slots[j] : 0;
        P.cy[j] = j < count ? cy[j] : 0.0;
        P.cs[j] = j < count ? cs[j] : 0.0;
    }
    const double bytes = 8.0 * (double)n * (2.0 + 2.0 * count);
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LBFGS_FINAL)) return rc;
    if (big) k_lbfgs_combine<true><<<grid, BLOCK, 0, st>>>(P);
    else k_lbfgs_combine<false><<<grid, BLOCK, 0, st>>>(P);
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_rows(ctx_, grid, NS)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s)) return rc;
    if (prof_on_) prof_commit(KK_LBFGS_FINAL, bytes);
    out.gu = s[S_GU]; out.uu = s[S_UU];
    return CGO_OK;
}

// The direction pass of the Gram form fused with phase 1 of the next line search's first trial (log-sum-exp objective:
// k_lbfgs_combine_lse).  CGO_LBFGS_FUSE_TRIAL=0 keeps the two launches (A/B).
bool HipBackend::lbfgs_direction_gram_can_fuse_trial() const {
    return pol_.lbfgs_fuse_trial != 0 && gram_on_ && (obj_->two_phase() || (spec_on_ && qn_m_ - 1 <= SPEC_MAXC));
}

int HipBackend::lbfgs_direction_gram_trial(const int *slots, const double *cy, const double *cs, int count, double cg, double a_trial,
                                           Scal &dir, Scal &trial) {
    HIPCHK(hipSetDevice(ctx_->device));
    if (count > GRAM_MAXC) { set_error("internal: Gram form limited to 12 pairs"); return CGO_EINVAL; }
    spec_valid_ = false;
    if (spec_on_ && count <= SPEC_MAXC && qn_m_ - 1 <= SPEC_MAXC) return lbfgs_direction_spec(slots, cy, cs, count, cg, a_trial, dir, trial);
    if (int rc = flush_lite()) return rc;
    if (!obj_->two_phase()) { set_error("internal: k_lbfgs_combine_lse is the log-sum-exp objective's"); return CGO_EINVAL; }
    const int64_t n = obj_->n_local;
    GramDirParams P;
    P.g = g_; P.u = u_.p; P.S = qn_S_.p; P.Y = qn_Y_.p; P.n = n; P.count = count; P.cg = cg;
    P.partials = ctx_->partials;
    for (int j = 0; j < GRAM_MAXC; ++j) {
        P.slots[j] = j < count ? slots[j] : 0;
        P.cy[j] = j < count ? cy[j] : 0.0;
        P.cs[j] = j < count ? cs[j] : 0.0;
    }
    const double bytes = 8.0 * (double)n * (3.0 + 2.0 * count);   // g, x, the ring / u
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LBFGS_FINAL)) return rc;
    if (big) k_lbfgs_combine_lse<true><<<grid, BLOCK, 0, st>>>(P, xc_, a_trial);
    else k_lbfgs_combine_lse<false><<<grid, BLOCK, 0, st>>>(P, xc_, a_trial);
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_launch(ctx_, grid, true)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s, MERGE_LSE)) return rc;
    if (prof_on_) prof_commit(KK_LBFGS_FINAL, bytes);
    dir.gu = s[S_GU]; dir.uu = s[S_UU];
    lse_a_ = a_trial; lse_M_ = s[L_M]; lse_S_ = s[L_S];
    lse_have_ = std::isfinite(lse_M_) && lse_S_ > 0.0 && std::isfinite(lse_S_);
    trial = Scal();
    trial.f = (s[L_M] + std::log(s[L_S])) + 0.5 * obj_->s0 * s[L_Q];   // as lse_stats: ϕ = lse + ½λ‖xp‖², dϕ = softmax·u + λ xp·u
    trial.gtu = s[L_T] / s[L_S] + obj_->s0 * s[L_R];
    return CGO_OK;
}

// ---- one ring pass per outer iteration: direction + first trial + every inner product of the NEXT push, taken at that trial
// (k_lbfgs_combine_spec; CGO_LBFGS_SPEC=0 keeps the two-pass form) ---------------------------------------------------------
template <class Obj>
static void launch_spec(bool big, bool push, int grid, hipStream_t st, const GramDirParams &P, const double *x, double a_trial, const SpecParams &Q, const SpecPush &U) {
    if (push) {
        if (big) k_lbfgs_combine_spec<Obj, true, true><<<grid, BLOCK, 0, st>>>(P, x, a_trial, Q, U);
        else k_lbfgs_combine_spec<Obj, false, true><<<grid, BLOCK, 0, st>>>(P, x, a_trial, Q, U);
    } else {
        if (big) k_lbfgs_combine_spec<Obj, true, false><<<grid, BLOCK, 0, st>>>(P, x, a_trial, Q, U);
        else k_lbfgs_combine_spec<Obj, false, false><<<grid, BLOCK, 0, st>>>(P, x, a_trial, Q, U);
    }
}
template <class Obj>
static void launch_lite(bool big, int grid, hipStream_t st, double *x, const double *u, double *g, double *sn, double *yn, const double *p0, long long n,
                        double a, double a_s, double M, double S, double lambda) {
    if (big) k_lbfgs_push_lite<Obj, true><<<grid, BLOCK, 0, st>>>(x, u, g, sn, yn, p0, n, a, a_s, M, S, lambda);
    else k_lbfgs_push_lite<Obj, false><<<grid, BLOCK, 0, st>>>(x, u, g, sn, yn, p0, n, a, a_s, M, S, lambda);
}

int HipBackend::lbfgs_direction_spec(const int *slots, const double *cy, const double *cs, int count, double cg, double a_trial,
                                     Scal &dir, Scal &trial) {
    const int64_t n = obj_->n_local;
    GramDirParams P;
    P.g = g_; P.u = u_.p; P.S = qn_S_.p; P.Y = qn_Y_.p; P.n = n; P.count = count; P.cg = cg;
    P.partials = ctx_->partials;
    for (int j = 0; j < GRAM_MAXC; ++j) {
        P.slots[j] = j < count ? slots[j] : 0;
        P.cy[j] = j < count ? cy[j] : 0.0;
        P.cs[j] = j < count ? cs[j] : 0.0;
    }
    // The reference: lse(x) itself, from the statistics of the current iterate (the last evaluated trial was accepted as x) —
    // then e_i = exp(xp_i − M_r) ≤ 1 at the maximum of x, S_r = Σ exp(x_i − M_r) = 1 up to rounding (κ = S_r/S' takes care of
    // the rest: ANY reference gives the same g⁺ = κ·p + λ·xp), and S' = exp(lse(xp) − lse(x)) is the change of the log-sum-exp
    // along the step.  The reference follows the iterate, whichever kernel produced its statistics.
    const bool lse = obj_->two_phase();
    const double Mr = lse ? lse_M_ + std::log(lse_S_) : 0.0, Sr = 1.0;
    SpecParams Q{Mr, 1.0 / Sr, obj_->s0, obj_->p0.p};
    const double hp = obj_->uses_param() ? 1.0 : 0.0;
    // the state update of the accepted speculated trial, if it was left to this pass (lbfgs_push_commit(direction_follows))
    const bool push = lite_deferred_;
    lite_deferred_ = false;
    SpecPush U{};
    if (push) {
        U.x = xc_; U.g = g_; U.a = lite_a_; U.a_s = lite_as_; U.M = lite_M_; U.S = lite_S_;
        U.sn = qn_S_.p + (size_t)lite_slot_ * (size_t)n; U.yn = qn_Y_.p + (size_t)lite_slot_ * (size_t)n;
        U.new_in_list = (count > 0 && slots[0] == lite_slot_) ? 1 : 0;   // (a pair with s·y ≤ 0 is written but does not join the history)
        push_counts_[0]++;
        qn_sgt_slot_ = -1;
    }
    // g, x, the ring / u — and with the state update: u_old / x, g, s, y, less the two reads of the pair formed in registers
    const double bytes = 8.0 * (double)n * (3.0 + hp + 2.0 * count + (push ? 5.0 - 2.0 * U.new_in_list : 0.0));
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LBFGS_FINAL)) return rc;
    switch (obj_->kind) {
    case CGO_OBJ_LSE: launch_spec<ObjLse>(big, push, grid, st, P, xc_, a_trial, Q, U); break;
    case CGO_OBJ_QUAD_DIAG: launch_spec<ObjQuadDiag>(big, push, grid, st, P, xc_, a_trial, Q, U); break;
    case CGO_OBJ_ROSENBROCK_PAIRED: launch_spec<ObjRosenPaired>(big, push, grid, st, P, xc_, a_trial, Q, U); break;
    case CGO_OBJ_USER: {   // the run-time compiled objective carries its own instantiations
        hipFunction_t f = obj_->rtc ? obj_->rtc->spec(big, push) : nullptr;
        if (!f) { set_error("internal: kernel missing from the run-time compiled objective module"); return CGO_EINVAL; }
        const double *xin = xc_;
        double at = a_trial;
        void *args[] = {(void *)&P, (void *)&xin, (void *)&at, (void *)&Q, (void *)&U};
        HIPCHK(hipModuleLaunchKernel(f, grid, 1, 1, BLOCK, 1, 1, 0, st, args, nullptr));
        break;
    }
    default: set_error("internal: no one-pass L-BFGS kernel for this objective"); return CGO_EINVAL;
    }
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_rows(ctx_, grid, NG)) return rc;
    double s[NG];
    if (int rc = fetch_sums(ctx_, s, MERGE_SUM, NG)) return rc;
    if (prof_on_) prof_commit(KK_LBFGS_FINAL, bytes);
    dir.gu = s[SP_GU]; dir.uu = s[SP_UU];
    if (!lse) {   // element-wise objective: the sums ARE the trial's and the next push's
        trial = Scal();
        trial.f = s[SE_F]; trial.gtu = s[SE_GTU]; trial.gtgt = s[SE_GTGT];
        std::memcpy(spec_s_, s, sizeof s);
        spec_a_ = a_trial; spec_count_ = count; spec_dphi_ = trial.gtu;
        for (int j = 0; j < count; ++j) spec_slots_[j] = slots[j];
        spec_valid_ = true;
        spec_unmat_ = true;    // g⁺ of this trial exists nowhere in memory (materialize() evaluates it again if somebody needs it)
        return CGO_OK;
    }
    // S' = Σ exp(xp − M_r) = exp(lse(xp) − lse(x)).  A first trial far out (overflow, or everything underflowing) is evaluated
    // the usual way instead — k_lse_stats works from the true maximum of xp — and nothing was speculated.
    const double Sp = s[SP_S];
    if (!(Sp >= 1e-280 && Sp <= 1e280) || !std::isfinite(s[SP_T])) {
        spec_refreshed_++;
        return lse_stats(0, 0, 0, a_trial, trial, false);
    }
    trial = Scal();
    trial.f = (Mr + std::log(Sp)) + 0.5 * obj_->s0 * s[SP_Q];   // ϕ = lse + ½λ‖xp‖², dϕ = softmax·u + λ xp·u (as lse_stats, reference M_r)
    trial.gtu = s[SP_T] / Sp + obj_->s0 * s[SP_R];
    std::memcpy(spec_s_, s, sizeof s);
    spec_Mr_ = Mr; spec_Sr_ = Sr; spec_a_ = a_trial; spec_count_ = count; spec_dphi_ = trial.gtu;
    for (int j = 0; j < count; ++j) spec_slots_[j] = slots[j];
    spec_valid_ = true;
    lse_a_ = a_trial; lse_M_ = Mr; lse_S_ = Sp;   // (M_r, S') describe xp as well as its own (max, Σ) would
    lse_have_ = true;
    return CGO_OK;
}

// The push for the step a_x from the sums the direction pass left, if a_x IS the step it speculated on: fills G (inner
// products with y-based entries: s_j·y, y_j·y, y_j·s — the caller adds its stored s_j·g, y_j·g), launches nothing; the
// state update itself (k_lbfgs_push_lite, or the next direction pass) is lbfgs_push_commit().  false = not available: take the usual path.
bool HipBackend::lbfgs_push_spec(double a_x, double a_s, int slot, const int *prev, int count, GramOut &G) {
    if (!spec_valid_ || push_pending_ || push_lite_pending_) return false;
    const bool lse = obj_->two_phase();
    if (std::memcmp(&a_x, &spec_a_, sizeof(double)) != 0 || (lse && std::memcmp(&a_x, &lse_a_, sizeof(double)) != 0)) return false;
    if (count != spec_count_) return false;
    for (int j = 0; j < count; ++j) if (prev[j] != spec_slots_[j]) return false;
    const double *s = spec_s_;
    if (!lse) {   // (spec_valid_ implies that no trial has been evaluated since the direction pass: trial() clears it)
        G.sy = a_s * s[SE_UY]; G.yy = s[SE_YY]; G.sgn = a_s * s[SE_GTU]; G.ygn = s[SE_YGT]; G.gtgt = s[SE_GTGT];
        bool ok = std::isfinite(G.sy) && std::isfinite(G.yy) && std::isfinite(G.sgn) && std::isfinite(G.ygn) && G.gtgt >= 1e-280 && G.gtgt <= 1e300;
        for (int j = 0; j < count; ++j) {
            const double *q = s + SP_PAIR + 5 * j;
            G.sjg[j] = q[0]; G.yjg[j] = q[1]; G.sjyn[j] = q[2]; G.yjyn[j] = q[3]; G.yjsn[j] = a_s * q[4];
            ok = ok && std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]) && std::isfinite(q[3]) && std::isfinite(q[4]);
        }
        if (!ok) return false;
        G.materialized = true; G.y_based = false;
        push_lite_pending_ = true;
        lite_a_ = a_x; lite_as_ = a_s; lite_slot_ = slot; lite_M_ = 0.0; lite_S_ = 1.0;
        spec_valid_ = false; spec_unmat_ = false;
        return true;
    }
    const double Sp = s[SP_S], lam = obj_->s0;
    const double kappa = spec_Sr_ / Sp, d = kappa - 1.0;                     // g⁺ = κ·p + λ·xp
    const double sup = s[SP_T] / spec_Sr_;                                    // Σ u·p
    const double E0 = s[SP_E0], E1 = s[SP_E0 + 1], E2 = s[SP_E0 + 2], E3 = s[SP_E0 + 3], E4 = s[SP_E0 + 4], E5 = s[SP_E0 + 5];
    G.sy = a_s * (E3 + d * sup);
    G.yy = E0 + 2.0 * d * E1 + d * d * E2;
    G.sgn = a_s * spec_dphi_;                                                 // s·g⁺ = a_s·(u·g⁺)
    G.ygn = kappa * E1 + lam * E5 + d * (kappa * E2 + lam * E4);
    // ‖g⁺‖², g⁺ = ĝ + d·p: from the element-wise small ĝ = p + λ·xp, not from Σp², Σp·xp, Σxp² (which cancel (‖p‖/‖g⁺‖)²-fold near a minimiser)
    G.gtgt = s[SP_GH2] + 2.0 * d * s[SP_GHP] + d * d * E2;
    bool ok = std::isfinite(kappa) && std::isfinite(G.sy) && std::isfinite(G.yy) && std::isfinite(G.ygn) &&
              G.gtgt >= 1e-280 && G.gtgt <= 1e300;   // (outside: the scaled-norm rare path wants a stored g⁺ — usual push)
    // y = ŷ + (κ − 1)·p is a sum of like-sized terms only while p does not dwarf y: p = S'·softmax(xp), so a step along which the
    // log-sum-exp RISES by more than log 2 (the ridge term paying for it) would have ŷ ≈ p ≫ y and the sums cancel S'-fold —
    // found by the seeded sweep (λ = 1e-6, iterates around −500: S' = 1e17, every y-sum came out 0).  Such a trial is as good a
    // trial as any (ϕ and dϕ are plain sums), but its push is the usual one.
    ok = ok && Sp <= 2.0;
    for (int j = 0; j < count; ++j) {
        const double *q = s + SP_PAIR + 5 * j;
        G.sjyn[j] = q[0] + d * q[1];
        G.yjyn[j] = q[2] + d * q[3];
        G.yjsn[j] = a_s * q[4];
        G.sjg[j] = G.yjg[j] = 0.0;
        ok = ok && std::isfinite(G.sjyn[j]) && std::isfinite(G.yjyn[j]) && std::isfinite(G.yjsn[j]);
    }
    if (!ok) return false;
    G.materialized = true; G.y_based = true;
    push_lite_pending_ = true;
    lite_a_ = a_x; lite_as_ = a_s; lite_slot_ = slot; lite_M_ = spec_Mr_; lite_S_ = Sp;
    spec_valid_ = false;
    return true;
}

int HipBackend::flush_lite() {
    if (!lite_deferred_) return CGO_OK;
    lite_deferred_ = false;
    return lbfgs_push_lite();
}

int HipBackend::lbfgs_push_lite() {
    HIPCHK(hipSetDevice(ctx_->device));
    const int64_t n = obj_->n_local;
    double *sn = qn_S_.p + (size_t)lite_slot_ * (size_t)n, *yn = qn_Y_.p + (size_t)lite_slot_ * (size_t)n;
    const double bytes = 8.0 * (double)n * (7.0 + (obj_->uses_param() ? 1.0 : 0.0));
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LBFGS_PUSH)) return rc;
    switch (obj_->kind) {
    case CGO_OBJ_LSE: launch_lite<ObjLse>(big, grid, st, xc_, u_.p, g_, sn, yn, obj_->p0.p, n, lite_a_, lite_as_, lite_M_, lite_S_, obj_->s0); break;
    case CGO_OBJ_QUAD_DIAG: launch_lite<ObjQuadDiag>(big, grid, st, xc_, u_.p, g_, sn, yn, obj_->p0.p, n, lite_a_, lite_as_, lite_M_, lite_S_, obj_->s0); break;
    case CGO_OBJ_ROSENBROCK_PAIRED: launch_lite<ObjRosenPaired>(big, grid, st, xc_, u_.p, g_, sn, yn, obj_->p0.p, n, lite_a_, lite_as_, lite_M_, lite_S_, obj_->s0); break;
    case CGO_OBJ_USER: {
        hipFunction_t f = obj_->rtc ? obj_->rtc->lite(big) : nullptr;
        if (!f) { set_error("internal: kernel missing from the run-time compiled objective module"); return CGO_EINVAL; }
        double *xa = xc_, *ga = g_, *sna = sn, *yna = yn;
        const double *ua = u_.p, *pa = obj_->p0.p;
        long long nn = n;
        double a = lite_a_, as = lite_as_, M = lite_M_, S = lite_S_, lam = obj_->s0;
        void *args[] = {&xa, &ua, &ga, &sna, &yna, &pa, &nn, &a, &as, &M, &S, &lam};
        HIPCHK(hipModuleLaunchKernel(f, grid, 1, 1, BLOCK, 1, 1, 0, st, args, nullptr));
        break;
    }
    default: set_error("internal: no one-pass L-BFGS kernel for this objective"); return CGO_EINVAL;
    }
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (prof_on_) prof_commit(KK_LBFGS_PUSH, bytes);
    qn_sgt_slot_ = -1;
    push_counts_[0]++;
    return CGO_OK;
}

// finalize + make the sums of the launch just enqueued available to the NEXT kernel on the
// device (dot_ptr) or, with a host communicator, on the host (dot_host).
int HipBackend::chain_sums(int grid, int slot, const double **dot_ptr, int *dot_count, double *dot_host) {
    hipStream_t st = ctx_->stream;
    if (int rc = finalize_rows(ctx_, grid, NS)) return rc;
    *dot_host = 0.0;
    if (ctx_->single()) { *dot_ptr = ctx_->out_dev; *dot_count = 1; return CGO_OK; }
    if (ctx_->shm()) {  // blocks live in host shared memory: one host round trip per step
        double sums[NS];
        if (int rc = fetch_sums(ctx_, sums)) return rc;
        *dot_ptr = nullptr; *dot_count = 0; *dot_host = sums[slot];
        return CGO_OK;
    }
    if (int rc = ctx_->ensure_gather()) return rc;
    const int dr = ctx_->comm->allgather_device(ctx_->out_dev, ctx_->gather_dev, NS, (void *)st);
    if (dr == 0) { *dot_ptr = ctx_->gather_dev; *dot_count = ctx_->world(); return CGO_OK; }
    if (dr > 0) return CGO_ECOMM;
    double sums[NS];
    if (int rc = fetch_sums(ctx_, sums)) return rc;  // host communicator: one round trip per step
    *dot_ptr = nullptr; *dot_count = 0; *dot_host = sums[slot];
    return CGO_OK;
}

int HipBackend::lbfgs_push(double a_x, double a_s, int slot, double &sy, double &yy) {
    HIPCHK(hipSetDevice(ctx_->device));
    const int64_t n = obj_->n_local;
    PushParams P;
    P.x = xc_; P.u = u_.p; P.g = g_; P.gt = gt_;
    P.s = qn_S_.p + (size_t)slot * (size_t)n; P.y = qn_Y_.p + (size_t)slot * (size_t)n;
    P.n = n; P.a = a_x; P.a_s = a_s; P.partials = ctx_->partials;
    const double bytes = 8.0 * (double)n * 7.0;
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    if (int rc = prof_begin(KK_LBFGS_PUSH)) return rc;
    if (big) k_lbfgs_push<true><<<grid, BLOCK, 0, ctx_->stream>>>(P);
    else k_lbfgs_push<false><<<grid, BLOCK, 0, ctx_->stream>>>(P);
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_rows(ctx_, grid, NS)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s)) return rc;
    if (prof_on_) prof_commit(KK_LBFGS_PUSH, bytes);
    sy = s[PS_SY]; yy = s[PS_YY];
    qn_sgt_ = s[PS_SGT];
    qn_sgt_slot_ = slot;      // Σ s_slot·g⁺ is the first dot of the two-loop if this pair is kept
    std::swap(g_, gt_);       // g ← g⁺
    return CGO_OK;
}

int HipBackend::lbfgs_direction(const int *slots, const double *rho, int count, double gamma, Scal &out) {
    HIPCHK(hipSetDevice(ctx_->device));
    if (count == 0) return reset_dir(out);  // no curvature pairs yet: u = −g
    const int64_t n = obj_->n_local;
    const double bytes = 8.0 * (double)n * 4.0;
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    auto S = [&](int slot) { return qn_S_.p + (size_t)slot * (size_t)n; };
    auto Y = [&](int slot) { return qn_Y_.p + (size_t)slot * (size_t)n; };
    LoopParams P;
    std::memset(&P, 0, sizeof(P));
    P.n = n; P.partials = ctx_->partials; P.alpha = qn_alpha_dev_; P.dot_stride = NS; P.dot_slot = S_GU;
    auto launch = [&](int kk, double nvec) -> int {
        if (int rc = prof_begin(kk)) return rc;
        if (big) k_lbfgs_loop<true><<<grid, BLOCK, 0, st>>>(P);
        else k_lbfgs_loop<false><<<grid, BLOCK, 0, st>>>(P);
        HIPCHK(hipGetLastError());
        if (int rc = prof_end()) return rc;
        total_launches_++;
        if (prof_on_) prof_commit(kk, 8.0 * (double)n * nvec);
        return CGO_OK;
    };
    // first dot  s_newest · g : already reduced by the push of this very pair, else one dot-only launch
    if (slots[0] == qn_sgt_slot_) {
        P.dot_ptr = nullptr; P.dot_count = 0; P.dot_host = qn_sgt_;
    } else {
        P.mode = 2; P.qin = g_; P.qout = u_.p; P.v = g_; P.w = S(slots[0]);
        if (int rc = launch(KK_LBFGS_LOOP, 2.0)) return rc;
        if (int rc = chain_sums(grid, S_GU, &P.dot_ptr, &P.dot_count, &P.dot_host)) return rc;
    }
    qn_sgt_slot_ = -1;
    for (int k = 0; k < count; ++k) {  // newest → oldest
        P.mode = 0; P.k = k; P.rho = rho[slots[k]];
        P.qin = (k == 0) ? g_ : u_.p; P.qout = u_.p; P.v = Y(slots[k]);
        P.apply_scale = (k == count - 1); P.scale = gamma; P.final_step = 0;
        P.w = (k < count - 1) ? S(slots[k + 1]) : Y(slots[count - 1]);
        if (int rc = launch(KK_LBFGS_LOOP, 4.0)) return rc;
        if (int rc = chain_sums(grid, S_GU, &P.dot_ptr, &P.dot_count, &P.dot_host)) return rc;
    }
    for (int k = count - 1; k >= 0; --k) {  // oldest → newest
        P.mode = 1; P.k = k; P.rho = rho[slots[k]];
        P.qin = u_.p; P.qout = u_.p; P.v = S(slots[k]);
        P.apply_scale = 0; P.final_step = (k == 0);
        P.w = (k > 0) ? Y(slots[k - 1]) : g_;
        if (int rc = launch(k == 0 ? KK_LBFGS_FINAL : KK_LBFGS_LOOP, 4.0)) return rc;
        if (k > 0) {
            if (int rc = chain_sums(grid, S_GU, &P.dot_ptr, &P.dot_count, &P.dot_host)) return rc;
        }
    }
    if (int rc = finalize_rows(ctx_, grid, NS)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s)) return rc;
    out.gu = s[S_GU]; out.uu = s[S_UU];
    return CGO_OK;
}

// LinearAlgebra.norm rare path: two tiny-output passes over one vector (16 B/elt in total)
int HipBackend::scaled_norm_parts(int which, double a_trial, double &maxabs, double &scaled_ss, bool &has_nan) {
    if (int rc = flush_lite()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    const int64_t n = obj_->n_local;
    const double *v = nullptr, *w = nullptr;   // the vector is v, or v − w
    if (which == 3) {                         // u is always stored
        if (int rc = pipe_drain()) return rc;
        v = rmode_ ? uc_ : u_.p;
    } else if (rmode_) {  // rare path: materialise the vector whose norm is asked for (g, or g⁺ of the last trial,
        const double a1[1] = {a_trial};  // or — which = 2, solvesystem — the gradient at the second iterate buffer)
        if (which == 2) std::swap(xc_, xn_);
        int rc = launch_r(KK_SCALED_NORM, (which == 1 || which == 4) ? R_GRADT : R_GRAD, 0, 0, a1, 1, false, nullptr);
        if (which == 2) std::swap(xc_, xn_);
        if (rc) return rc;
        v = ga_.p;
        if (which == 4) {   // y = g⁺ − g: g⁺ sits in ga_ now; g = ∇f(x) goes to a scratch buffer
            if (sys_on_) { set_error("internal: scaled norm of y is not available to solvesystem"); return CGO_EINVAL; }
            if ((rc = ensure_gb())) return rc;
            std::swap(ga_.p, gb_.p);   // R_GRAD writes to ga_: let it write into the scratch, then swap back
            rc = launch_r(KK_SCALED_NORM, R_GRAD, 0, 0, a1, 1, false, nullptr);
            std::swap(ga_.p, gb_.p);
            if (rc) return rc;
            v = ga_.p; w = gb_.p;
        }
    } else if (which == 2) {
        set_error("internal: scaled norm of the second iterate needs the k_cg family");
        return CGO_EINVAL;
    } else if (which == 4) {
        v = gt_; w = g_;
    } else {
        v = which ? gt_ : g_;
    }
    hipStream_t st = ctx_->stream;
    int grid = (int)std::min<int64_t>((n + BLOCK - 1) / BLOCK, GRID_SMALL);
    if (grid < 1) grid = 1;
    double s[NS];
    double *hp; unsigned long long *hs;
    k_scaled_norm<0><<<grid, BLOCK, 0, st>>>(v, w, n, 1.0, ctx_->partials);
    HIPCHK(hipGetLastError());
    ctx_->seq++;
    ctx_->pub_target(&hp, &hs);
    k_finalize_maxsum<0><<<1, 64, 0, st>>>(ctx_->partials, grid, ctx_->out_dev, hp, hs, ctx_->seq);
    HIPCHK(hipGetLastError());
    total_launches_++;
    if (int rc = fetch_sums(ctx_, s, MERGE_MAX0)) return rc;
    maxabs = s[0]; has_nan = s[1] > 0.0; scaled_ss = 0.0;
    if (has_nan || maxabs == 0.0 || std::isinf(maxabs)) return CGO_OK;
    k_scaled_norm<1><<<grid, BLOCK, 0, st>>>(v, w, n, maxabs, ctx_->partials);
    HIPCHK(hipGetLastError());
    ctx_->seq++;
    ctx_->pub_target(&hp, &hs);
    k_finalize_maxsum<1><<<1, 64, 0, st>>>(ctx_->partials, grid, ctx_->out_dev, hp, hs, ctx_->seq);
    HIPCHK(hipGetLastError());
    total_launches_++;
    if (int rc = fetch_sums(ctx_, s)) return rc;
    scaled_ss = s[0];
    if (prof_on_) { prof_cnt_[KK_SCALED_NORM] += 2; prof_bytes_[KK_SCALED_NORM] = 8.0 * (double)n; }
    return CGO_OK;
}

// ---- resident solver (cgo_resident.hpp, cgo_kernels_resident.hip.hpp) -------------------------------------------------
// Which shards: the built-in element-wise objectives under a CG β and one of the two bisection line searches, on one rank,
// while x, u (and the parameter vector) fit the LDS of the chip's CUs — one workgroup per CU at most, so that every
// workgroup of the launch is resident and their all-gather can complete.  CGO_RESIDENT=0 switches it off,
// CGO_RES_CHUNK sets the elements per workgroup (default 4096: n = 1e6 → 245 workgroups; n ≤ 4096 → ONE workgroup and no
// exchange at all), CGO_RES_POINTS the trial steps per pass (default 3).
constexpr int64_t RES_REC_CAP = 4096;     // iterations per slice at most
constexpr int64_t RES_LOG_CAP = 1 << 16;  // trial-log entries per slice

template <class Obj>
static const void *res_kernel(int npts) {
    return npts >= 7 ? (const void *)k_resident<Obj, 7> : (npts >= 3 ? (const void *)k_resident<Obj, 3> : (const void *)k_resident<Obj, 1>);
}
static const void *res_kernel_for(int obj_kind, int npts) {
    switch (obj_kind) {
    case CGO_OBJ_ROSENBROCK_CHAINED: return npts >= 3 ? (const void *)k_resident_chain<3> : (const void *)k_resident_chain<1>;   // ONE workgroup
    case CGO_OBJ_QUAD_DIAG: return res_kernel<ObjQuadDiag>(npts);
    case CGO_OBJ_ROSENBROCK_PAIRED: return res_kernel<ObjRosenPaired>(npts);
    case CGO_OBJ_BOOTH: return res_kernel<ObjBooth>(npts);
    default: return nullptr;
    }
}

int HipBackend::res_plan() {
    if (res_grid_ != 0) return res_grid_ > 0 ? res_grid_ : 0;
    res_grid_ = -1;   // decided: does not fit, unless the plan below completes
    const int64_t want = pol_.resident_chunk >= 2 ? (int64_t)(pol_.resident_chunk & ~1) : (int64_t)4096;
    const int pts = (pol_.resident_points == 1 || pol_.resident_points == 3 || pol_.resident_points == 7) ? pol_.resident_points : 3;
    res_npts_ = pts;
    const void *fn = res_kernel_for(obj_->kind, res_npts_);
    // a run-time compiled objective carries its own copy of the kernel (k_resident<UserObjective, 3>, cgo_rtc.hip)
    hipFunction_t mf = (obj_->kind == CGO_OBJ_USER && obj_->rtc) ? obj_->rtc->resident(res_npts_) : nullptr;
    if (!fn && !mf) return 0;
    const int64_t n = obj_->n_local;
    const int vecs = chain() ? 4 : (obj_->uses_param() ? 3 : 2);   // (the stencil objective: two LDS copies of x and of u)
    if (chain() && res_npts_ > 3) res_npts_ = 3;
    int max_lds = 0;
    if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx_->device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int64_t static_lds = 0;
    if (fn) {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, fn) != hipSuccess) { (void)hipGetLastError(); return 0; }
        static_lds = (int64_t)fa.sharedSizeBytes;
    } else {
        int v = 0;
        if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, mf) != hipSuccess) { (void)hipGetLastError(); return 0; }
        static_lds = v;
    }
    const int64_t avail = (int64_t)max_lds - static_lds - 512;
    int64_t chunk_max = (avail / (8 * vecs)) & ~1LL;
    if (chunk_max < 2) return 0;
    const int cus = std::min(ctx_->num_cu > 0 ? ctx_->num_cu : 256, RES_GSIZE * RES_GROUPS);   // (the two-level exchange holds 16 groups of 16)
    int64_t chunk = std::min<int64_t>(want, chunk_max);
    if (chain()) {   // the whole (padded) vector in ONE workgroup, or not at all
        chunk = n + (n & 1);
        if (chunk > chunk_max) return 0;
    }
    int64_t grid = (n + chunk - 1) / chunk;
    if (grid > cus) {   // more elements per workgroup, up to what the LDS holds
        chunk = (((n + cus - 1) / cus) + 1) & ~1LL;
        if (chunk > chunk_max) return 0;
        grid = (n + chunk - 1) / chunk;
    }
    const size_t lds = (size_t)chunk * 8 * vecs;
    if (fn && lds > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int per_cu = 0;
    if (fn) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, BLOCK, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); return 0; } }
    else if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mf, BLOCK, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); return 0; }
    if (grid > (int64_t)cus * per_cu) return 0;   // every workgroup must be resident: they wait for one another
    res_chunk_ = chunk; res_lds_ = lds; res_grid_ = (int)grid;
    return res_grid_;
}

bool HipBackend::resident_ready(const cgo_cg_config &cfg, const cgo_ls_config &ls) const {
    if (!res_on_ || !rmode_ || sys_on_ || !ctx_->single()) return false;
    if (cfg.beta.kind == CGO_BETA_LBFGS) return false;
    if (ls.kind != CGO_LS_STRONG_WOLFE_BISECTION && ls.kind != CGO_LS_WOLFE_BISECTION) return false;
    return const_cast<HipBackend *>(this)->res_plan() > 0;
}

int HipBackend::res_alloc() {
    if (res_state_) return CGO_OK;
    HIPCHK(hipSetDevice(ctx_->device));
    HIPCHK(hipHostMalloc((void **)&res_state_, sizeof(ResState), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&res_recs_, sizeof(ResRecord) * RES_REC_CAP, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&res_done_, 64, hipHostMallocDefault));
    *res_done_ = 0;
    const size_t xb = sizeof(double) * RES_XBUFS * ((size_t)res_grid_ + RES_GROUPS) * RES_WMAX;   // workgroup rows, then group rows
    HIPCHK(hipMalloc((void **)&res_xbuf_, xb));
    HIPCHK(hipMemsetD32((hipDeviceptr_t)res_xbuf_, (int)(TAIL_EMPTY & 0xFFFFFFFFull), xb / 4));
    HIPCHK(hipMalloc((void **)&res_recs_dev_, sizeof(ResRecord) * RES_REC_CAP));
    HIPCHK(hipMalloc((void **)&res_err_, 64));     // [0] error flags, [1] workgroups that have reported in
    HIPCHK(hipMemset(res_err_, 0, 64));
    if (res_grid_ > 1) {   // a multi-workgroup slice leaves x, u in these; swapped in on a good global verdict only
        if (int rc = res_xo_.alloc((size_t)obj_->n_local)) return rc;
        if (int rc = res_uo_.alloc((size_t)obj_->n_local)) return rc;
    }
    HIPCHK(hipDeviceSynchronize());
    res_round_ = 0;
    return CGO_OK;
}

int HipBackend::resident_run(const ResConfig &c, ResState &s, int64_t budget, std::vector<ResRecord> &recs, std::vector<ResLog> &log) {
    if (int rc = pipe_drain()) return rc;
    pipe_streak_ = 0;
    if (res_plan() <= 0) { set_error("internal: resident slice on a shard that does not fit"); return CGO_ESTATE; }
    if (obj_->uses_param() && !obj_->p0_set) { set_error("objective parameter vector (slot 0) was never set"); return CGO_ESTATE; }
    if (int rc = res_alloc()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    if (c.log_on && !res_log_) {
        HIPCHK(hipHostMalloc((void **)&res_log_, sizeof(ResLog) * RES_LOG_CAP, hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&res_log_dev_, sizeof(ResLog) * RES_LOG_CAP));
    }
    ResParams P{};
    P.x = xc_; P.u = uc_; P.p0 = obj_->p0.p; P.n = obj_->n_local; P.chunk = res_chunk_; P.s0 = obj_->s0;
    const bool oop = res_grid_ > 1;
    double *xo = oop ? ((xc_ == res_xo_.p) ? res_xin_ : res_xo_.p) : xc_, *uo = oop ? ((uc_ == res_uo_.p) ? res_uin_ : res_uo_.p) : uc_;
    P.xo = xo; P.uo = uo; P.arrive = res_err_ + 1;
    P.inject = -1;
    if (res_slices_ == 0) { if (const char *e = getenv("CGO_RES_INJECT_GIVEUP")) P.inject = atoi(e); }   // test hook: first slice only
    P.cfg = c; P.cfg.npts = res_npts_;
    P.st = s;
    if (P.st.ncache > res_npts_) P.st.ncache = res_npts_;   // (a wider host launch left more trial results than a pass of this width keeps)
    P.budget = std::min<int64_t>(budget, RES_REC_CAP);
    P.st_out = res_state_; P.recs = res_recs_dev_; P.log = res_log_dev_; P.log_cap = c.log_on ? RES_LOG_CAP : 0;
    P.recs_host = res_recs_; P.log_host = res_log_;
    P.xbuf = res_xbuf_; P.round0 = res_round_; P.err = res_err_;
    P.done_seq = res_done_; P.seq = ++res_seq_;
    static const bool timing = getenv("CGO_RES_TIMING") != nullptr;
    P.timing = timing ? 1 : 0;
    const void *fn = res_kernel_for(obj_->kind, res_npts_);
    void *args[] = {&P};
    const double h0 = timing ? now_ns() : 0.0;
    if (int rc = prof_begin(KK_RESIDENT)) return rc;
    if (fn) HIPCHK(hipLaunchKernel(fn, dim3(res_grid_), dim3(BLOCK), args, res_lds_, ctx_->stream));
    else HIPCHK(hipModuleLaunchKernel(obj_->rtc->resident(res_npts_), res_grid_, 1, 1, BLOCK, 1, 1, (unsigned)res_lds_, ctx_->stream, args, nullptr));
    if (int rc = prof_end()) return rc;
    total_launches_++;
    const double h1 = timing ? now_ns() : 0.0;
    if (int rc = wait_word(ctx_, res_done_, res_seq_)) return rc;
    const double h2 = timing ? now_ns() : 0.0;
    if (timing) fprintf(stderr, "[cgo resident] host: enqueue %.1f us, wait for the slice %.1f us\n", (h1 - h0) * 1e-3, (h2 - h1) * 1e-3);
    s = *res_state_;
    res_round_ += (unsigned long long)s.passes;
    res_slices_++;
    {
        if (timing) fprintf(stderr, "[cgo resident] slice: %lld iterations, %lld passes, grid %d x %lld elements, reason %d: %.1f us in all; per pass compute %.2f, "
                                 "workgroup reduce %.2f, exchange %.2f us; outside the passes %.2f us per iteration (machine %.2f, evals incl. passes %.2f, post %.2f); shader clock %.0f MHz\n",
                         (long long)s.done, (long long)s.passes, res_grid_, (long long)res_chunk_, (int)s.reason, s.t_total * 1e-2,
                         s.t_compute * 1e-2 / std::max<double>(s.passes, 1), s.t_reduce * 1e-2 / std::max<double>(s.passes, 1),
                         s.t_exchange * 1e-2 / std::max<double>(s.passes, 1),
                         (s.t_total - s.t_compute - s.t_reduce - s.t_exchange) * 1e-2 / std::max<double>(s.done, 1),
                         s.t_machine * 1e-2 / std::max<double>(s.done, 1), s.t_eval * 1e-2 / std::max<double>(s.done, 1), s.t_post * 1e-2 / std::max<double>(s.done, 1),
                         (double)s.t_cycles / std::max<double>((double)s.t_total * 1e-2, 1e-9));
    }
    if (s.reason == RES_ERROR) {
        // The exchange gave up: some workgroup of the launch was not running while the others waited for its row.  That
        // happens when ANOTHER process's kernels hold CUs (two persistent launches can each be partially resident and wait
        // for workgroups the other one's keep out).  Nothing is lost: a slice writes x, u back only when it ends well, so
        // the state is still that of the slice's start — hand the whole slice to the launch-per-trial engine and keep this
        // solver off the resident path from here on (correct under any sharing of the GPU, at the old speed).
        HIPCHK(hipStreamSynchronize(ctx_->stream));
        const size_t xb = sizeof(double) * RES_XBUFS * ((size_t)res_grid_ + RES_GROUPS) * RES_WMAX;
        HIPCHK(hipMemsetD32((hipDeviceptr_t)res_xbuf_, (int)(TAIL_EMPTY & 0xFFFFFFFFull), xb / 4));
        HIPCHK(hipMemset(res_err_, 0, 64));
        res_round_ = 0;
        res_on_ = false;
        res_gave_up_++;
        s = P.st;   // the state the slice started from
        s.done = 0; s.log_len = 0; s.evals = 0; s.passes = 0; s.reason = RES_HOST;
        recs.clear(); log.clear();
        return CGO_OK;
    }
    if (oop && s.done > 0) {   // a good slice, by the verdict of ALL its workgroups: its x, u become the iterate
        res_xin_ = xc_; res_uin_ = uc_;
        xc_ = xo; uc_ = uo;
    }
    res_iters_ += s.done;
    recs.assign(res_recs_, res_recs_ + s.done);
    if (c.log_on) log.assign(res_log_, res_log_ + s.log_len); else log.clear();
    // state moved once per slice: load x, u (+ D) and store x, u
    if (prof_on_) prof_commit(KK_RESIDENT, 8.0 * (double)obj_->n_local * (double)((obj_->uses_param() ? 3 : 2) + (s.done > 0 ? 2 : 0)));
    return CGO_OK;
}

// Did a finisher of a fused launch ever give up on a partial row (finish_tail's bounded poll)?  Its sums carry a NaN then;
// asked for with the results so that such a solve ends in an error, not in a status that blames the objective.
int HipBackend::tail_errors() {
    if (!ctx_->fused_tail) return CGO_OK;
    HIPCHK(hipSetDevice(ctx_->device));
    unsigned int e = 0;   // (ordered behind everything enqueued on the stream, armed rounds included)
    HIPCHK(hipMemcpyAsync(&e, ctx_->tickets + TAIL_GROUP + 1, sizeof e, hipMemcpyDeviceToHost, ctx_->stream));
    HIPCHK(hipStreamSynchronize(ctx_->stream));
    if (e) {   // report once, then start over from clean mailboxes: the next solve on this context is not poisoned by this one
        const unsigned int zero = 0;
        (void)hipMemcpyAsync(ctx_->tickets + TAIL_GROUP + 1, &zero, sizeof zero, hipMemcpyHostToDevice, ctx_->stream);
        (void)hipMemsetD32Async((hipDeviceptr_t)ctx_->partials_f, (int)(TAIL_EMPTY & 0xFFFFFFFFull), (size_t)MAX_GRID * NR7 * 2, ctx_->stream);
        (void)hipMemsetD32Async((hipDeviceptr_t)ctx_->partials2_f, (int)(TAIL_EMPTY & 0xFFFFFFFFull), (size_t)TAIL_GROUP * NG * 2, ctx_->stream);
        (void)hipMemsetAsync(ctx_->tickets, 0, sizeof(unsigned int) * (TAIL_GROUP + 1), ctx_->stream);
        (void)hipStreamSynchronize(ctx_->stream);
        set_error("a launch's reduction tail gave up waiting for " + std::to_string(e) + " partial-row slot(s): the sums of this solve are not trustworthy");
        return CGO_EHIP;
    }
    return CGO_OK;
}

int HipBackend::download(double *x, double *g) {
    if (int rc = flush_lite()) return rc;
    if (int rc = pipe_drain()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    if (rmode_ && g) {  // the gradient lives only in registers during the solve: materialise ∇f(x) now
        if (int rc = launch_r(KK_INIT, R_GRAD, 0, 0, nullptr, 0, false, nullptr)) return rc;
        g_ = ga_.p;
    }
    const size_t nb = sizeof(double) * (size_t)obj_->n_local;
    if (x) HIPCHK(hipMemcpyAsync(x, xc_, nb, hipMemcpyDeviceToHost, ctx_->stream));
    if (g) HIPCHK(hipMemcpyAsync(g, g_, nb, hipMemcpyDeviceToHost, ctx_->stream));
    HIPCHK(hipStreamSynchronize(ctx_->stream));
    return CGO_OK;
}

int HipBackend::download_device(double *x_dev, double *g_dev) {
    if (int rc = flush_lite()) return rc;
    if (int rc = pipe_drain()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    if (rmode_ && g_dev) {  // the gradient lives only in registers during the solve: materialise ∇f(x) now
        if (int rc = launch_r(KK_INIT, R_GRAD, 0, 0, nullptr, 0, false, nullptr)) return rc;
        g_ = ga_.p;
    }
    const size_t nb = sizeof(double) * (size_t)obj_->n_local;
    if (x_dev) HIPCHK(hipMemcpyAsync(x_dev, xc_, nb, hipMemcpyDeviceToDevice, ctx_->stream));
    if (g_dev) HIPCHK(hipMemcpyAsync(g_dev, g_, nb, hipMemcpyDeviceToDevice, ctx_->stream));
    HIPCHK(hipStreamSynchronize(ctx_->stream));
    return CGO_OK;
}

// ---------------------------------------------------------------- raw single-launch helpers
namespace {
struct Tmp {  // host vector → device copy
    DevBuf b;
    int up(HipCtx *c, const double *h, int64_t n) {
        if (int rc = b.alloc((size_t)n)) return rc;
        if (h) HIPCHK(hipMemcpyAsync(b.p, h, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        return CGO_OK;
    }
    int down(HipCtx *c, double *h, int64_t n) {
        HIPCHK(hipMemcpyAsync(h, b.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        return CGO_OK;
    }
};
KParams base_params(HipCtx *c, int64_t n) {
    KParams P;
    std::memset(&P, 0, sizeof(P));
    P.n = n;
    P.partials = c->partials; P.out = c->out_dev;
    return P;
}
}  // namespace

int HipBackend::run_dir(HipCtx *ctx, double *u, const double *g, double beta, int64_t n, double *out2) {
    HIPCHK(hipSetDevice(ctx->device));
    Tmp du, dg;
    if (int rc = du.up(ctx, u, n)) return rc;
    if (int rc = dg.up(ctx, g, n)) return rc;
    KParams P = base_params(ctx, n);
    P.u = du.b.p; P.g = dg.b.p; P.beta = beta;
    if (int rc = launch_fused(ctx, 0, M_DIR, &P, n)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx, s)) return rc;
    if (int rc = du.down(ctx, u, n)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    out2[0] = s[S_GU]; out2[1] = s[S_UU];
    return CGO_OK;
}

int HipBackend::run_beta_partials(HipCtx *ctx, const double *gn, const double *g, const double *u,
                                  int64_t n, double *out9) {
    HIPCHK(hipSetDevice(ctx->device));
    Tmp dgn, dg, du;
    if (int rc = dgn.up(ctx, gn, n)) return rc;
    if (int rc = dg.up(ctx, g, n)) return rc;
    if (int rc = du.up(ctx, u, n)) return rc;
    KParams P = base_params(ctx, n);
    P.gt = dgn.b.p; P.g = dg.b.p; P.u = du.b.p;
    if (int rc = launch_fused(ctx, 0, M_BETAONLY, &P, n)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx, s)) return rc;
    out9[0] = s[S_GTU]; out9[1] = s[S_GTGT]; out9[2] = s[S_GTG]; out9[3] = s[S_YY];
    out9[4] = s[S_UY]; out9[5] = s[S_YGT]; out9[6] = s[S_GG]; out9[7] = s[S_GU]; out9[8] = s[S_UU];
    return CGO_OK;
}

int HipBackend::run_trial(HipObjective *obj, const double *x, const double *u, double a,
                          double *gn_out, double *out2) {
    HipCtx *ctx = obj->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n = obj->n_local;
    if (obj->two_phase()) {  // phase 1 (ϕ, dϕ) then phase 2 (g⁺) on a scratch state
        HipBackend b(ctx, obj);
        if (int rc = b.alloc()) return rc;
        if (int rc = b.set_x0_host(x)) return rc;
        HIPCHK(hipMemcpyAsync(b.u_.p, u, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        b.need_beta_ = false;
        Scal s;
        if (int rc = b.lse_stats(0, 0, 0, a, s, false)) return rc;
        out2[0] = s.f; out2[1] = s.gtu;
        if (gn_out) {
            if (int rc = b.lse_grad(false, a, s)) return rc;
            HIPCHK(hipMemcpyAsync(gn_out, b.gt_, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
        return CGO_OK;
    }
    Tmp dx, du, dgt;
    if (int rc = dx.up(ctx, x, n)) return rc;
    if (int rc = du.up(ctx, u, n)) return rc;
    if (int rc = dgt.up(ctx, nullptr, n)) return rc;
    KParams P = base_params(ctx, n);
    P.x = dx.b.p; P.u = du.b.p; P.gt = dgt.b.p; P.p0 = obj->p0.p; P.a_trial = a; P.s0 = obj->s0;
    P.offset = obj->offset;
    if (int rc = launch_fused(ctx, obj->kind, M_TRIAL, &P, n, false, obj)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx, s)) return rc;
    if (gn_out) {
        if (int rc = dgt.down(ctx, gn_out, n)) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    out2[0] = s[S_F]; out2[1] = s[S_GTU];
    return CGO_OK;
}

int HipBackend::run_eval(HipObjective *obj, const double *x, double *g_out, double *f) {
    HipCtx *ctx = obj->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n = obj->n_local;
    if (obj->two_phase() || obj->kind == CGO_OBJ_ROSENBROCK_CHAINED) {   // through a scratch solver state
        HipBackend b(ctx, obj);
        if (obj->kind == CGO_OBJ_ROSENBROCK_CHAINED) { b.set_rmode(true); b.set_multi_min_n(INT64_MAX); }
        if (int rc = b.alloc()) return rc;
        if (int rc = b.set_x0_host(x)) return rc;
        Scal s;
        if (int rc = b.init_eval(s)) return rc;
        *f = s.f;
        return g_out ? b.download(nullptr, g_out) : CGO_OK;
    }
    Tmp dx, du, dgt;
    if (int rc = dx.up(ctx, x, n)) return rc;
    if (int rc = du.up(ctx, nullptr, n)) return rc;
    if (int rc = dgt.up(ctx, nullptr, n)) return rc;
    KParams P = base_params(ctx, n);
    P.x = dx.b.p; P.u = du.b.p; P.gt = dgt.b.p; P.p0 = obj->p0.p; P.s0 = obj->s0;
    P.offset = obj->offset;
    if (int rc = launch_fused(ctx, obj->kind, M_INIT, &P, n, false, obj)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx, s)) return rc;
    if (g_out) {
        if (int rc = dgt.down(ctx, g_out, n)) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    *f = s[S_F];
    return CGO_OK;
}

// The read/write mix of the dominant launch without its arithmetic: median and best of `reps` launches (HIP events).
int HipBackend::bench_stream_mix(HipCtx *ctx, int64_t n, int reps, double *median_us, double *best_us) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n < 2 || reps < 1 || reps > 1000) { set_error("bench_stream_mix: n ≥ 2 and 1 ≤ reps ≤ 1000 required"); return CGO_EINVAL; }
    DevBuf x, u, d;
    if (int rc = x.alloc((size_t)n)) return rc;
    if (int rc = u.alloc((size_t)n)) return rc;
    if (int rc = d.alloc((size_t)n)) return rc;
    const int fg = (int)std::min<int64_t>((n + BLOCK - 1) / BLOCK, GRID_SMALL);
    hipStream_t st = ctx->stream;
    k_fill<<<fg, BLOCK, 0, st>>>(x.p, n, 0, 1, 1, -1.0, 1.0);
    k_fill<<<fg, BLOCK, 0, st>>>(u.p, n, 0, 1, 2, -1.0, 1.0);
    k_fill<<<fg, BLOCK, 0, st>>>(d.p, n, 0, 1, 3, 1.0, 10.0);
    std::vector<float> t((size_t)reps);
    for (int r = -2; r < reps; ++r) {
        if (r >= 0) HIPCHK(hipEventRecord(ctx->ev0, st));
        k_stream_mix<true><<<GRID_BIG, BLOCK, 0, st>>>(x.p, u.p, d.p, n, 1e-9, 0.5);
        if (r >= 0) {
            HIPCHK(hipEventRecord(ctx->ev1, st));
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipEventElapsedTime(&t[(size_t)r], ctx->ev0, ctx->ev1));
        }
    }
    HIPCHK(hipGetLastError());
    std::sort(t.begin(), t.end());
    *median_us = (double)t[(size_t)reps / 2] * 1e3;
    *best_us = (double)t[0] * 1e3;
    return CGO_OK;
}

// Device-resident micro-benchmark of one fused kernel kind (no host traffic in
// the timed region): mean HIP-event time over `reps` back-to-back launches.
int HipBackend::bench_kernel(HipCtx *ctx, HipObjective *obj, int kernel_kind, int64_t n, int reps,
                             double *ms, double *bytes) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n < 2 || reps < 1) { set_error("bench_kernel: n ≥ 2 and reps ≥ 1 required"); return CGO_EINVAL; }
    int mode = 0;
    switch (kernel_kind) {
    case KK_INIT: mode = M_INIT; break;
    case KK_TRIAL: mode = M_TRIAL | M_BETA; break;
    case KK_ACCEPT_DIR_TRIAL: mode = M_ACCEPT | M_DIR | M_TRIAL | M_BETA; break;
    case KK_ACCEPT_DIR: mode = M_ACCEPT | M_DIR; break;
    case KK_ACCEPT_ONLY: mode = M_ACCEPT; break;
    case KK_RESET_DIR: mode = M_RESET; break;
    case KK_UPG_NORM: mode = M_UPG; break;
    case 100: mode = M_DIR; break;       // the 24 B/elt updatedir!+dots kernel
    case 101: mode = M_BETAONLY; break;  // the 24 B/elt getβ partial-sum kernel
    default: set_error("bench_kernel: unknown kernel kind"); return CGO_EINVAL;
    }
    const int okind = obj ? obj->kind : CGO_OBJ_ROSENBROCK_PAIRED;
    if (obj && obj->uses_param() && (!obj->p0_set || obj->n_local < n)) {
        set_error("bench_kernel: objective parameter vector missing or shorter than n");
        return CGO_EINVAL;
    }
    DevBuf x, u, g, gt;
    if (int rc = x.alloc((size_t)n)) return rc;
    if (int rc = u.alloc((size_t)n)) return rc;
    if (int rc = g.alloc((size_t)n)) return rc;
    if (int rc = gt.alloc((size_t)n)) return rc;
    const int fg = (int)std::min<int64_t>((n + BLOCK - 1) / BLOCK, GRID_SMALL);
    k_fill<<<fg, BLOCK, 0, ctx->stream>>>(x.p, n, 0, 1, 1, -1.0, 1.0);
    k_fill<<<fg, BLOCK, 0, ctx->stream>>>(u.p, n, 0, 1, 2, -1.0, 1.0);
    k_fill<<<fg, BLOCK, 0, ctx->stream>>>(g.p, n, 0, 1, 3, -1.0, 1.0);
    k_fill<<<fg, BLOCK, 0, ctx->stream>>>(gt.p, n, 0, 1, 4, -1.0, 1.0);
    KParams P = base_params(ctx, n);
    P.x = x.p; P.u = u.p; P.g = g.p; P.gt = gt.p; P.p0 = obj ? obj->p0.p : nullptr;
    P.s0 = obj ? obj->s0 : 0.0;
    P.a_acc = 1e-9; P.beta = 0.5; P.a_trial = 1e-3;  // keeps values bounded over many reps
    for (int w = 0; w < 2; ++w)
        if (int rc = launch_fused(ctx, okind, mode, &P, n, false, obj)) return rc;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < reps; ++r)
        if (int rc = launch_fused(ctx, okind, mode, &P, n, false, obj)) return rc;
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
    *ms = (double)t / reps;
    *bytes = bytes_for(okind, mode, n, obj && obj->uses_param());
    return CGO_OK;
}

}  // namespace cgo

#ifdef CGO_STAMPS
// diagnostic build: the per-workgroup stamps of the last k_cg launch (cgo_kernels_cg.hip.hpp); the caller has synchronised
extern "C" int cgo_debug_stamps(unsigned long long *out, int words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cgo::dev::cgo_stamps), (size_t)words * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
