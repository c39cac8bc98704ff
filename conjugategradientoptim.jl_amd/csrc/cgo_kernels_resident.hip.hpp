// cgo_kernels_resident.hip.hpp — k_resident: a slice of WHOLE outer iterations in one launch, state resident in LDS.
//
// BASELINE configs 1 and 2 (n = 1e3, 1e6) are latency-bound under a launch per trial: 1–15 µs of vector work, then a kernel
// boundary, a PCIe publish, the host's decision and the next launch (3.5 launches per iteration on config 1,
// 17.5 k it/s against 44–80 k on ONE CPU thread; config 2 re-read 24 MB from L2 / Infinity Cache per launch).  Here:
//
//   * each of ≤ 256 workgroups (one per CU) loads its contiguous chunk of x, u and the parameter vector into LDS ONCE
//     (≤ 160 KB: 6 600 elements with a parameter vector, 9 900 without) and writes x, u back when the slice ends;
//   * every thread of every workgroup runs res_iterate (cgo_resident.hpp) — the reference's outer loop, its line search,
//     getβ, the direction update — on identical global sums: replicated control flow, no broadcast of decisions;
//   * a pass = the fused bodies of cgo_kernels_cg.hip.hpp (cg_pair: same expressions, same per-element bits) over the
//     chunk in LDS + the transpose-reduce + ONE all-gather of a row of sums per workgroup:
//       - a row slot is a self-validating 8-byte granule: TAIL_EMPTY (a signalling NaN no sum can produce) until its value
//         arrives; written with one sc1 (agent-scope) store, polled with sc1 loads — the "data-tagged granule" hand-off
//         of MI355X_MICROARCH.md (allgather row: 256 producers → every CU ≈ 3 µs) — no flag, no fence, no atomics;
//       - four rotating row buffers: in round r a workgroup publishes into buffer r mod 4 and clears ITS OWN row of buffer
//         (r + 2) mod 4 — everybody finished reading that one before publishing round r − 1, which this workgroup has
//         seen complete — and drains its stores (s_waitcnt vmcnt(0)) before it leaves the round, so a cleared slot is
//         EMPTY a full round before anyone polls it;
//       - rows are summed in workgroup order by every workgroup alike ⇒ bitwise identical sums everywhere, run to run;
//   * the polls are bounded: a workgroup that never sees a row gives up, raises the error word and ends; so do the others.
//
// One record per completed iteration (trace) and the trial log go straight to pinned host memory from workgroup 0.
#pragma once

#include "cgo_kernels_cg.hip.hpp"
#ifndef CGO_RTC
#include "cgo_kernels_chain.hip.hpp"
#endif
#include "cgo_resident.hpp"

namespace cgo {
namespace dev {

constexpr int RES_XBUFS = 4;
constexpr int RES_GSIZE = 16;            // workgroups per group of the two-level exchange
constexpr int RES_GROUPS = 16;           // groups at most (256 workgroups)
constexpr int RES_WMAX = NR7;            // row stride of the exchange buffers (widest row)
constexpr int RES_SPIN = 1 << 19;        // polls of one slot before giving up (≈ a second)

struct ResParams {
    double *x; double *u; const double *p0;   // this rank's shard in HBM
    double *xo; double *uo;                   // where a slice that ends well leaves x, u: OTHER buffers when the launch has more than one
                                              // workgroup (the host swaps them in only on a GLOBAL verdict), = x, u for one workgroup
    unsigned int *arrive;                     // device: workgroups 1 … that have finished (their error flags are final)
    int inject;                               // test hook (CGO_RES_INJECT_GIVEUP): this workgroup behaves as if its poll gave up in the slice's LAST pass; −1 off
    long long n, chunk;                       // elements; elements per workgroup (even)
    double s0;
    ResConfig cfg;
    ResState st;                              // the state the slice starts from (by value: scalar loads)
    long long budget;
    ResState *st_out;                         // pinned host
    ResRecord *recs;                          // DEVICE [budget]: the leader's stores must not wait for PCIe; copied out when the slice ends
    ResLog *log; long long log_cap;           // DEVICE
    ResRecord *recs_host; ResLog *log_host;   // pinned host
    double *xbuf;                             // [RES_XBUFS][grid][RES_WMAX] workgroup rows + [RES_XBUFS][RES_GROUPS][RES_WMAX] group rows; TAIL_EMPTY wherever no value is in flight
    unsigned long long round0;                // exchange round this launch starts at
    unsigned int *err;                        // device: bumped when a poll gave up
    int timing;                               // CGO_RES_TIMING=1: read the clock around the phases of every pass
    unsigned long long *done_seq; unsigned long long seq;   // pinned: released by workgroup 0 once st_out / recs / log are complete
};

template <class Obj, int NPTS>
struct ResDev {
    const ResParams &P;
    double *xs, *us, *ps;      // LDS
    int npairs; bool odd;      // pairs of this chunk; does this workgroup own the odd tail element (local index 2·npairs)?
    unsigned long long round;
    double *tot;               // LDS [RES_WMAX]
    double *fs;                // LDS [BLOCK]
    long long t_compute = 0, t_reduce = 0, t_exchange = 0;   // 100 MHz ticks (wall_clock64) spent in the three phases of a pass

    static constexpr int kNpts = NPTS;
    __device__ __forceinline__ bool leader() const { return blockIdx.x == 0 && threadIdx.x == 0; }
    // (a clock read is an s_memrealtime round trip of ≈ 0.2 µs: with ≈ 30 of them per outer iteration the instrumentation
    //  itself cost a third of a small slice — only with CGO_RES_TIMING=1)
    __device__ __forceinline__ long long clock() const { return P.timing ? wall_clock64() : 0; }

    // One row of sums per workgroup → the same totals in every workgroup (tot[]), in TWO hops:
    //   1. every workgroup publishes its row; the first workgroup of each group of 16 polls its group's rows and adds them
    //      in workgroup order → the group's row;
    //   2. every workgroup polls the ≤ 16 group rows and adds them in group order.
    // A flat all-gather (every workgroup reads all 245 rows) was built first: 47 KB per workgroup and round, 11.5 MB over the
    // chip — 6.0 µs per round at n = 1e6 with 24-slot rows (4.8 µs with 10-slot rows), bandwidth- not latency-bound.  Two hops
    // move 2 × 3 KB per workgroup.  Slots are self-validating granules (TAIL_EMPTY until the value lands), rows rotate through
    // four buffers and are cleared by their owner two rounds ahead (see the header of this file) — for both levels alike.
    template <int W>
    __device__ __forceinline__ int poll_rows(const unsigned long long *rows, int nrows, double &t) {
        // lane (g, sl) of the first 16·W ≤ … lanes: rows g, g + Gp, … of slot sl, added in row order
        int bad = 0;
        const int tid = threadIdx.x;
        constexpr int Gp = BLOCK / W;
        t = 0.0;
        if (tid < Gp * W) {
            const int g = tid / W, sl = tid - g * W;
            unsigned long long b[4];   // ≤ 16 rows over Gp ≥ 4 lanes-groups: at most four rows per lane
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = g + k * Gp;
                b[k] = (r < nrows) ? __hip_atomic_load(rows + (size_t)r * RES_WMAX + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = g + k * Gp;
                for (int spin = 0; b[k] == TAIL_EMPTY && spin < RES_SPIN; ++spin) {
                    __builtin_amdgcn_s_sleep(1);
                    b[k] = __hip_atomic_load(rows + (size_t)r * RES_WMAX + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((spin & 1023) == 1023 && __hip_atomic_load(P.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // somebody gave up: so do we
                }
                if (b[k] == TAIL_EMPTY) bad = 1;
                t += __longlong_as_double((long long)b[k]);   // (+0.0 for rows past the end changes nothing)
            }
        }
        return bad;
    }

    template <int W>
    __device__ __forceinline__ int exchange(double own) {
        static_assert(BLOCK / W >= 4, "poll_rows keeps at most four rows per lane");
        const int G = gridDim.x, tid = threadIdx.x;
        if (G == 1) {
            if (tid < W) tot[tid] = own;
            __syncthreads();
            return 0;
        }
        constexpr int Gp = BLOCK / W;
        const int grp = blockIdx.x / RES_GSIZE, ngroups = (G + RES_GSIZE - 1) / RES_GSIZE;
        const bool head = (blockIdx.x % RES_GSIZE) == 0;
        unsigned long long *base = reinterpret_cast<unsigned long long *>(P.xbuf);
        const size_t lvl2 = (size_t)RES_XBUFS * G * RES_WMAX;   // the group rows live behind the workgroup rows
        unsigned long long *rows1 = base + (size_t)(round % RES_XBUFS) * G * RES_WMAX;
        unsigned long long *clr1 = base + (size_t)((round + 2) % RES_XBUFS) * G * RES_WMAX;
        unsigned long long *rows2 = base + lvl2 + (size_t)(round % RES_XBUFS) * RES_GROUPS * RES_WMAX;
        unsigned long long *clr2 = base + lvl2 + (size_t)((round + 2) % RES_XBUFS) * RES_GROUPS * RES_WMAX;
        if (tid < RES_WMAX) __hip_atomic_store(clr1 + (size_t)blockIdx.x * RES_WMAX + tid, TAIL_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (head && tid < RES_WMAX) __hip_atomic_store(clr2 + (size_t)grp * RES_WMAX + tid, TAIL_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < W) __hip_atomic_store(rows1 + (size_t)blockIdx.x * RES_WMAX + tid, (unsigned long long)__double_as_longlong(own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int bad = 0;
        if (head) {   // hop 1: this group's rows → the group's row
            const int left = G - grp * RES_GSIZE, in_group = left < RES_GSIZE ? left : RES_GSIZE;
            double t;
            bad = poll_rows<W>(rows1 + (size_t)grp * RES_GSIZE * RES_WMAX, in_group, t);
            if (tid < Gp * W) fs[tid] = t;
            __syncthreads();
            if (tid < W) {
                double r = 0.0;
#pragma unroll
                for (int g = 0; g < Gp; ++g) r += fs[g * W + tid];
                __hip_atomic_store(rows2 + (size_t)grp * RES_WMAX + tid, (unsigned long long)__double_as_longlong(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();   // fs is written again below
        }
        {   // hop 2: the group rows → the totals
            double t;
            bad |= poll_rows<W>(rows2, ngroups, t);
            if (tid < Gp * W) fs[tid] = t;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this round's clears and publishes have reached the fabric
        bad = __syncthreads_or(bad);
        if (bad) {
            if (tid == 0) atomicAdd(P.err, 1u);
            return 9;
        }
        if (tid < W) {
            double r = 0.0;
#pragma unroll
            for (int g = 0; g < Gp; ++g) r += fs[g * W + tid];
            tot[tid] = r;
        }
        __syncthreads();
        ++round;
        return 0;
    }

    // one pass over the chunk in LDS: the fused body of mode MODE with NP trial points, the workgroup's row, the exchange
    template <int MODE, int NP>
    __device__ __forceinline__ int pass(double a_acc, double beta, const double *a, int k, double (&sums)[RW<NP>::W]) {
        constexpr int W = RW<NP>::W;
        const int tid = threadIdx.x;
        RParams rp;
        rp.a_acc = a_acc; rp.beta = beta; rp.s0 = P.s0; rp.n = 0;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) rp.a[j] = (j < NP && k > 0) ? a[j] : 0.0;   // the caller pads a[] to NP entries
        double acc[W];
#pragma unroll
        for (int s = 0; s < W; ++s) acc[s] = 0.0;
        const long long t0 = clock();
        d2 *x2 = reinterpret_cast<d2 *>(xs), *u2 = reinterpret_cast<d2 *>(us);
        const d2 *p2 = reinterpret_cast<const d2 *>(ps);
        int i = tid;
        for (; i + BLOCK < npairs; i += 2 * BLOCK) {     // two independent pairs per trip
            d2 xa = x2[i], xb = x2[i + BLOCK], ua = u2[i], ub = u2[i + BLOCK];
            const d2 pa = Obj::kParam ? p2[i] : d2{0.0, 0.0}, pb = Obj::kParam ? p2[i + BLOCK] : d2{0.0, 0.0};
            bool wxa = false, wua = false, wxb = false, wub = false;
            d2 ga, gb;
            cg_pair<Obj, MODE, NP>(rp, xa, ua, pa, acc, wxa, wua, ga);
            cg_pair<Obj, MODE, NP>(rp, xb, ub, pb, acc, wxb, wub, gb);
            if (wxa) x2[i] = xa;
            if (wua) u2[i] = ua;
            if (wxb) x2[i + BLOCK] = xb;
            if (wub) u2[i + BLOCK] = ub;
        }
        if (i < npairs) {
            d2 xa = x2[i], ua = u2[i];
            const d2 pa = Obj::kParam ? p2[i] : d2{0.0, 0.0};
            bool wxa = false, wua = false;
            d2 ga;
            cg_pair<Obj, MODE, NP>(rp, xa, ua, pa, acc, wxa, wua, ga);
            if (wxa) x2[i] = xa;
            if (wua) u2[i] = ua;
        }
        if (odd && tid == 0) {   // the odd tail element of the global vector (objectives that are not pair-only)
            rp.x = xs; rp.u = us; rp.p0 = ps; rp.xo = xs; rp.uo = us; rp.gout = nullptr; rp.x2 = nullptr;
            cg_single<Obj, MODE, NP>(rp, 2LL * npairs, acc);
        }
        const long long t1 = clock();
        const double own = wg_reduce_n<W>(acc);
        const long long t2 = clock();
        if (int rc = exchange<W>(own)) return rc;
#pragma unroll
        for (int s = 0; s < W; ++s) sums[s] = tot[s];
        __syncthreads();   // tot and fs are written again by the next pass
        t_compute += t1 - t0; t_reduce += t2 - t1; t_exchange += clock() - t2;
        return 0;
    }

    static __device__ TrialSums ts(const double *q) { return TrialSums{q[0], q[1], q[2], q[3], q[4], q[5], q[6]}; }

    __device__ __forceinline__ int trial(const double *a, int k, TrialSums *out) {
        constexpr int NT = NPTS < 3 ? NPTS : 3;
        double sums[RW<NT>::W];
        if (int rc = pass<R_TRIAL, NT>(0.0, 0.0, a, k, sums)) return rc;
#pragma unroll
        for (int j = 0; j < NT; ++j) out[j] = ts(sums + RS_PER_POINT * j);   // (all NT: a padded point repeats a real one)
        return 0;
    }
    __device__ __forceinline__ int accept_dir_trial(double a_acc, double beta, const double *a, int k, TrialSums *out, double &gu, double &uu) {
        if (k == 0) {
            double sums[RW<1>::W];
            if (int rc = pass<R_ACCEPT | R_DIR, 1>(a_acc, beta, a, 0, sums)) return rc;
            gu = sums[RW<1>::GU]; uu = sums[RW<1>::UU];
            return 0;
        }
        double sums[RW<NPTS>::W];
        if (int rc = pass<R_ACCEPT | R_DIR | R_TRIAL, NPTS>(a_acc, beta, a, k, sums)) return rc;
#pragma unroll
        for (int j = 0; j < NPTS; ++j) out[j] = ts(sums + RS_PER_POINT * j);
        gu = sums[RW<NPTS>::GU]; uu = sums[RW<NPTS>::UU];
        return 0;
    }
};

template <class Obj, int NPTS>
__global__ __launch_bounds__(BLOCK, 1) void k_resident(const ResParams P) {   // one wave per SIMD is all a CU ever holds of this kernel: up to 512 VGPRs
    extern __shared__ __attribute__((aligned(16))) double res_lds[];
    __shared__ double tot[RES_WMAX];
    __shared__ double fs[BLOCK];
    __shared__ ResState s_out;
    const int tid = threadIdx.x;
    const long long lo = (long long)blockIdx.x * P.chunk;
    long long cnt = P.n - lo;
    if (cnt > P.chunk) cnt = P.chunk;
    if (cnt < 0) cnt = 0;
    double *xs = res_lds, *us = res_lds + P.chunk, *ps = res_lds + 2 * P.chunk;
    for (long long i = tid; i < cnt; i += BLOCK) {
        xs[i] = P.x[lo + i];
        us[i] = P.u[lo + i];
        if (Obj::kParam) ps[i] = P.p0[lo + i];
    }
    __syncthreads();
    ResDev<Obj, NPTS> v{P, xs, us, ps, (int)(cnt >> 1), (cnt & 1) != 0, P.round0, tot, fs, 0, 0, 0};
    // The loop state arrives as kernel arguments, i.e. in SGPRs, and the compiler would keep every value it can prove uniform
    // there: ≈ 160 + 30 scalar registers of state and configuration against 102 available — 300–900 SGPR spills, each reload a
    // v_readlane plus hazard wait states on the critical path of the scalar logic.  All of it is FP64 arithmetic anyway
    // (no scalar FP64 ALU): move the doubles to VGPRs once, opaquely.
    ResState s = P.st;
    ResConfig cfg = P.cfg;
#define RES_V(x) asm volatile("" : "+v"(x))
    RES_V(s.f_x); RES_V(s.gg); RES_V(s.norm); RES_V(s.dphi0); RES_V(s.uu); RES_V(s.a_initial); RES_V(s.last_a); RES_V(s.last_beta);
#pragma unroll
    for (int j = 0; j < RES_MAXP; ++j) {
        RES_V(s.ca[j]); RES_V(s.cs[j].f); RES_V(s.cs[j].gtu); RES_V(s.cs[j].gtgt); RES_V(s.cs[j].gtg); RES_V(s.cs[j].yy); RES_V(s.cs[j].uy); RES_V(s.cs[j].ygt);
    }
    RES_V(cfg.ls.c1); RES_V(cfg.ls.c2); RES_V(cfg.ls.a_max_growth_factor); RES_V(cfg.ls.delta1); RES_V(cfg.ls.max_step_size);
    RES_V(cfg.ls.discount_factor); RES_V(cfg.eps); RES_V(cfg.mu);
#undef RES_V
    const long long t_begin = wall_clock64(), c_begin = clock64();
    res_iterate(cfg, s, v, (int64_t)P.budget, P.recs, P.log, (int64_t)P.log_cap);
    s.t_cycles = clock64() - c_begin;
    s.t_total = wall_clock64() - t_begin; s.t_compute = v.t_compute; s.t_reduce = v.t_reduce; s.t_exchange = v.t_exchange;
    __syncthreads();
    // The slice's verdict must be GLOBAL before anything the host trusts is changed: the reason is decided per
    // workgroup, and a workgroup whose poll gave up in the slice's last pass can leave its peers none the wiser — they would
    // write their chunks back while it does not, and workgroup 0 would report a good slice over a mixed x, u.  So (1) a
    // launch of more than one workgroup writes x, u to OTHER buffers, which the host swaps in only for a good slice — a bad one
    // leaves the slice-start state untouched by construction; (2) every other workgroup reports in (after its last possible
    // error flag), and workgroup 0 publishes only once all have — bounded wait — and turns any error flag into RES_ERROR.
    if (P.inject >= 0 && (int)blockIdx.x == P.inject) {   // (test hook: the peers have completed the slice and know nothing)
        if (tid == 0) atomicAdd(P.err, 1u);
        s.reason = RES_ERROR;
    }
    if (s.done > 0 && s.reason != RES_ERROR) {   // x, u of the last completed iteration (an iteration handed back never touched them)
        for (long long i = tid; i < cnt; i += BLOCK) {
            P.xo[lo + i] = xs[i];
            P.uo[lo + i] = us[i];
        }
    }
    if (gridDim.x > 1) {
        if (blockIdx.x != 0) {
            __syncthreads();   // every lane's share of the write-back is issued before lane 0 reports in
            if (tid == 0) { __threadfence(); atomicAdd(P.arrive, 1u); }
        } else if (tid == 0) {
            unsigned got = 0;
            for (int spin = 0; spin < RES_SPIN; ++spin) {
                got = __hip_atomic_load(P.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (got == gridDim.x - 1) break;
                __builtin_amdgcn_s_sleep(2);
            }
            const unsigned e = __hip_atomic_load(P.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got != gridDim.x - 1 || e != 0u) s.reason = RES_ERROR;   // (the host clears both words on this path)
            else __hip_atomic_store(P.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (blockIdx.x == 0) {
        // records and trial log: device → pinned host, all lanes (the leader wrote them with plain stores from this CU)
        {
            const long long nr = s.done * (long long)(sizeof(ResRecord) / 8), nl = s.log_len * (long long)(sizeof(ResLog) / 8);
            const unsigned long long *rs = reinterpret_cast<const unsigned long long *>(P.recs), *ls = reinterpret_cast<const unsigned long long *>(P.log);
            unsigned long long *rd = reinterpret_cast<unsigned long long *>(P.recs_host), *ld = reinterpret_cast<unsigned long long *>(P.log_host);
            for (long long i = tid; i < nr; i += BLOCK) rd[i] = __hip_atomic_load(rs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (long long i = tid; i < nl; i += BLOCK) ld[i] = __hip_atomic_load(ls + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        constexpr int WS = sizeof(ResState) / 8;
        static_assert(sizeof(ResState) % 8 == 0 && WS <= BLOCK, "the state goes out as 8-byte words, one lane each");
        if (tid == 0) s_out = s;
        __syncthreads();
        if (tid < WS) reinterpret_cast<unsigned long long *>(P.st_out)[tid] = reinterpret_cast<const unsigned long long *>(&s_out)[tid];
        __threadfence_system();   // records, log and state before the word (once per slice: its price does not matter here)
        __syncthreads();
        if (tid == 0) __hip_atomic_store(P.done_seq, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#ifndef CGO_RTC
// ---- the stencil objective (chained Rosenbrock, cgo_kernels_chain.hip.hpp) in resident form: ONE workgroup ---------------------
// BASELINE config 1 in its chained form (n = 1000) is a launch-latency problem like the paired form.  ∇f_k reads x_{k±1}, so a
// pass reads the neighbouring pairs too — inside one workgroup that is just LDS (no halo, no exchange): x and u live in two
// LDS copies each, a pass reads one and writes the other (the launch-per-trial kernels do the same with two HBM buffers), the
// barriers of the workgroup reduction separate a pass's reads from the next pass's writes.  Up to ≈ 4 800 elements (four
// arrays in 160 KB); larger stencil problems keep their launches.  Same window arithmetic (chain_window), same row layout.
template <int NPTS>
struct ResDevChain {
    const ResParams &P;
    double *xa, *ua, *xb, *ub;   // LDS: current (a) and other (b) copies of x and u, padded to even length
    int npairs, odd;             // pairs incl. the padded one; odd: the last element is padding
    double *tot;
    long long t_compute = 0, t_reduce = 0, t_exchange = 0;
    static constexpr int kNpts = NPTS;
    __device__ __forceinline__ bool leader() const { return threadIdx.x == 0; }
    __device__ __forceinline__ long long clock() const { return P.timing ? wall_clock64() : 0; }

    template <int MODE, int NP>
    __device__ __forceinline__ int pass(double a_acc, double beta, const double *a, int k, double (&sums)[RW<NP>::W]) {
        constexpr int W = RW<NP>::W;
        constexpr bool wr_x = (MODE & R_ACCEPT) != 0, wr_u = (MODE & (R_DIR | R_INIT | R_RESET)) != 0;
        const int tid = threadIdx.x;
        const double a3[3] = {k > 0 ? a[0] : 0.0, (NP >= 3 && k > 0) ? a[1] : 0.0, (NP >= 3 && k > 0) ? a[2] : 0.0};   // the caller pads to NP entries
        double acc[W];
#pragma unroll
        for (int s = 0; s < W; ++s) acc[s] = 0.0;
        const long long t0 = clock();
        const d2 *x2 = reinterpret_cast<const d2 *>(xa), *u2 = reinterpret_cast<const d2 *>(ua);
        d2 *xo2 = reinterpret_cast<d2 *>(xb), *uo2 = reinterpret_cast<d2 *>(ub);
        for (int i = tid; i < npairs; i += BLOCK) {
            const bool first = (i == 0), last = (i == npairs - 1);
            double X[6], U[6];
            bool E[6];
            const d2 c = x2[i], cu = u2[i];
            const d2 l = first ? d2{0.0, 0.0} : x2[i - 1], lu = first ? d2{0.0, 0.0} : u2[i - 1];
            const d2 r = last ? d2{0.0, 0.0} : x2[i + 1], ru = last ? d2{0.0, 0.0} : u2[i + 1];
            X[0] = l.x; X[1] = l.y; X[2] = c.x; X[3] = c.y; X[4] = r.x; X[5] = r.y;
            U[0] = lu.x; U[1] = lu.y; U[2] = cu.x; U[3] = cu.y; U[4] = ru.x; U[5] = ru.y;
            E[0] = E[1] = !first;
            E[2] = true;
            E[3] = !(last && odd != 0);
            E[4] = E[5] = !last;
            if (odd != 0 && i + 1 == npairs - 1) E[5] = false;
            d2 xn2, un2, g2;
            chain_window<MODE, NP, W>(X, U, E, a_acc, beta, a3, acc, xn2, un2, g2);
            if (wr_x) xo2[i] = xn2;
            if (wr_u) uo2[i] = un2;
        }
        const long long t1 = clock();
        const double own = wg_reduce_n<W>(acc);   // (its barriers: every lane has finished reading the current copies)
        const long long t2 = clock();
        if (tid < W) tot[tid] = own;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < W; ++s) sums[s] = tot[s];
        __syncthreads();
        if (wr_x) { double *t = xa; xa = xb; xb = t; }
        if (wr_u) { double *t = ua; ua = ub; ub = t; }
        t_compute += t1 - t0; t_reduce += t2 - t1; t_exchange += clock() - t2;
        return 0;
    }
    static __device__ TrialSums ts(const double *q) { return TrialSums{q[0], q[1], q[2], q[3], q[4], q[5], q[6]}; }
    __device__ __forceinline__ int trial(const double *a, int k, TrialSums *out) {
        constexpr int NT = NPTS < 3 ? NPTS : 3;
        double sums[RW<NT>::W];
        if (int rc = pass<R_TRIAL, NT>(0.0, 0.0, a, k, sums)) return rc;
#pragma unroll
        for (int j = 0; j < NT; ++j) out[j] = ts(sums + RS_PER_POINT * j);
        return 0;
    }
    __device__ __forceinline__ int accept_dir_trial(double a_acc, double beta, const double *a, int k, TrialSums *out, double &gu, double &uu) {
        if (k == 0) {
            double sums[RW<1>::W];
            if (int rc = pass<R_ACCEPT | R_DIR, 1>(a_acc, beta, a, 0, sums)) return rc;
            gu = sums[RW<1>::GU]; uu = sums[RW<1>::UU];
            return 0;
        }
        double sums[RW<NPTS>::W];
        if (int rc = pass<R_ACCEPT | R_DIR | R_TRIAL, NPTS>(a_acc, beta, a, k, sums)) return rc;
#pragma unroll
        for (int j = 0; j < NPTS; ++j) out[j] = ts(sums + RS_PER_POINT * j);
        gu = sums[RW<NPTS>::GU]; uu = sums[RW<NPTS>::UU];
        return 0;
    }
};

template <int NPTS>
__global__ __launch_bounds__(BLOCK, 1) void k_resident_chain(const ResParams P) {
    static_assert(NPTS == 1 || NPTS == 3, "the stencil passes carry one or three trial points");
    extern __shared__ __attribute__((aligned(16))) double res_lds[];
    __shared__ double tot[RES_WMAX];
    __shared__ ResState s_out;
    const int tid = threadIdx.x;
    const long long npad = P.chunk;   // padded (even) length: the whole vector lives in this ONE workgroup
    double *xa = res_lds, *ua = res_lds + npad, *xb = res_lds + 2 * npad, *ub = res_lds + 3 * npad;
    for (long long i = tid; i < npad; i += BLOCK) {
        const bool real = i < P.n;
        xa[i] = real ? P.x[i] : 0.0; ua[i] = real ? P.u[i] : 0.0;
        xb[i] = 0.0; ub[i] = 0.0;
    }
    __syncthreads();
    ResDevChain<NPTS> v{P, xa, ua, xb, ub, (int)(npad >> 1), (int)(P.n & 1), tot, 0, 0, 0};
    ResState s = P.st;
    ResConfig cfg = P.cfg;
#define RES_V(x) asm volatile("" : "+v"(x))
    RES_V(s.f_x); RES_V(s.gg); RES_V(s.norm); RES_V(s.dphi0); RES_V(s.uu); RES_V(s.a_initial); RES_V(s.last_a); RES_V(s.last_beta);
#pragma unroll
    for (int j = 0; j < RES_MAXP; ++j) {
        RES_V(s.ca[j]); RES_V(s.cs[j].f); RES_V(s.cs[j].gtu); RES_V(s.cs[j].gtgt); RES_V(s.cs[j].gtg); RES_V(s.cs[j].yy); RES_V(s.cs[j].uy); RES_V(s.cs[j].ygt);
    }
    RES_V(cfg.ls.c1); RES_V(cfg.ls.c2); RES_V(cfg.ls.a_max_growth_factor); RES_V(cfg.ls.delta1); RES_V(cfg.ls.max_step_size);
    RES_V(cfg.ls.discount_factor); RES_V(cfg.eps); RES_V(cfg.mu);
#undef RES_V
    const long long t_begin = wall_clock64(), c_begin = clock64();
    res_iterate(cfg, s, v, (int64_t)P.budget, P.recs, P.log, (int64_t)P.log_cap);
    s.t_cycles = clock64() - c_begin;
    s.t_total = wall_clock64() - t_begin; s.t_compute = v.t_compute; s.t_reduce = v.t_reduce; s.t_exchange = v.t_exchange;
    __syncthreads();
    if (s.done > 0 && s.reason != RES_ERROR) {
        for (long long i = tid; i < P.n; i += BLOCK) { P.xo[i] = v.xa[i]; P.uo[i] = v.ua[i]; }   // (one workgroup: its verdict IS global; xo = x)
    }
    {
        const long long nr = s.done * (long long)(sizeof(ResRecord) / 8), nl = s.log_len * (long long)(sizeof(ResLog) / 8);
        const unsigned long long *rs = reinterpret_cast<const unsigned long long *>(P.recs), *ls = reinterpret_cast<const unsigned long long *>(P.log);
        unsigned long long *rd = reinterpret_cast<unsigned long long *>(P.recs_host), *ld = reinterpret_cast<unsigned long long *>(P.log_host);
        for (long long i = tid; i < nr; i += BLOCK) rd[i] = __hip_atomic_load(rs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (long long i = tid; i < nl; i += BLOCK) ld[i] = __hip_atomic_load(ls + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        constexpr int WS = sizeof(ResState) / 8;
        if (tid == 0) s_out = s;
        __syncthreads();
        if (tid < WS) reinterpret_cast<unsigned long long *>(P.st_out)[tid] = reinterpret_cast<const unsigned long long *>(&s_out)[tid];
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(P.done_seq, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#endif   // !CGO_RTC

}  // namespace dev
}  // namespace cgo
