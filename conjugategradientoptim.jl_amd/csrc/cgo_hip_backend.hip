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
#include "cgo_backend_internal.hpp"

#include "cgo_kernels.hip.hpp"
#include "cgo_kernels_lse.hip.hpp"
#include "cgo_kernels_cg.hip.hpp"
#include "cgo_kernels_chain.hip.hpp"

namespace cgo {

using namespace dev;

// How long a host wait on a publishing kernel or on a peer's mailbox slot may last before the solve is given up
// (a peer died, a collective hung): CGO_WAIT_TIMEOUT_S, default 120 s.
static double wait_timeout_ns() {
    static const double t = [] { const char *e = getenv("CGO_WAIT_TIMEOUT_S"); double v = e ? atof(e) : 0.0; return (v > 0.0 ? v : 120.0) * 1e9; }();
    return t;
}

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const char *get_error() { return g_err.c_str(); }


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
double big_bytes_for(double forced, bool read_only) {
    if (forced > 0.0) return forced;
    return read_only ? 4.5e8 : 1.4e9;
}
double env_big_bytes() {
    static const double forced = [] { const char *e = getenv("CGO_BIG_BYTES"); double t = e ? atof(e) : 0.0; return t > 0.0 ? t : 0.0; }();
    return forced;
}
double HipBackend::big_bytes(bool read_only) const { return big_bytes_for(pol_.hbm_stream_bytes, read_only); }
bool is_big(int obj_kind, int mode, int64_t n, bool hp, double forced) {
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
int grid_cg(int64_t n, int npts) {
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
int launch_module(hipFunction_t f, void *params, int grid, hipStream_t st) {
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
static int wait_seq(HipCtx *ctx, unsigned long long want) { return wait_word(ctx, ctx->host_seq, want); }
int wait_word(HipCtx *ctx, unsigned long long *word, unsigned long long want) {
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
int wait_checked(HipCtx *ctx, unsigned long long *word, unsigned long long want, const double *block, int ns, double *dst) {
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
