// cgo_kernels_cg.hip.hpp — gradient-free, multi-point CG kernels for element-wise objectives.
//
// Two MI355X-first ideas on top of k_fused (cgo_kernels.hip.hpp), both trading plentiful FP64
// VALU flops for HBM bytes — the only scarce resource of this path:
//
//  1. NO GRADIENT VECTOR IN HBM.  For an element-wise objective ∇f_i depends on x_i (and its
//     pair partner) only, so the current gradient g = ∇f(x) is recomputed in registers from the
//     x the launch reads anyway, instead of being stored by one launch and re-read by the next.
//     The recomputation is bit-identical to what the accepting trial computed: same expression,
//     same inputs (x ← x + a*·u is evaluated once, stored, and re-read), no FMA contraction.
//       accept+dir+trial :  R x,u,D ; W x,u        40 B/elt   (k_fused: R x,u,g,D ; W x,u,g⁺ = 56)
//       trial            :  R x,u,D ; W —          24 B/elt   (k_fused: R x,u,g,D ; W g⁺     = 40)
//     The reference moves (224 + 56k)·n bytes per outer iteration for the same arithmetic
//     (SURVEY.md §3.6).  The gradient is materialised once, when results are requested.
//
//  2. SPECULATIVE MULTI-POINT TRIALS.  The bisection line searches of the reference
//     (nocedal.jl:33-209, wolfe.jl:13-207, geometric.jl:102-152) pick the next step from at most
//     TWO candidates fixed by the bracket state *before* the current trial's outcome is known
//     (zoom midpoint vs extrapolation; lower vs upper half).  One launch therefore evaluates the
//     requested step AND both candidates (NPTS = 3; 5 and 7 add one / two more tree levels along the
//     two likeliest paths): ϕ, dϕ and every getβ partial sum for all
//     points from one pass over x,u,D.  The host state machine then walks two levels of the
//     decision tree per launch; the step sequence, and hence parity, is unchanged.
//
// Row layout: point j → 7j + {F, GTU, GTGT, GTG, YY, UY, YGT}; then g·u_new, u_new·u_new (direction
// part) at 7·NPTS, 7·NPTS+1; padded to 10 slots (NPTS = 1), 24 (NPTS = 3), 40 (NPTS = 5: the
// requested step, both candidates, and the likelier grandchild under each candidate) or 56
// (NPTS = 7: one more level along the same two paths).
#pragma once

#include "cgo_kernels.hip.hpp"
#ifndef CGO_RTC   // the run-time compiled copies of these kernels (user objectives) carry no controller
#include "cgo_ctl.hpp"
#endif

namespace cgo {
namespace dev {

constexpr int NR = 24;   // row width of 3-point launches
constexpr int NR1 = 10;  // row width of 1-point launches (= NS: shares k_finalize)
constexpr int NR5 = 40;  // row width of 5-point launches (35 trial sums + 2 direction sums, padded)
constexpr int NR7 = 56;  // row width of 7-point launches (49 + 2, padded)
constexpr int MAXP = 7;  // most trial points one launch evaluates
enum RSlot : int { RS_F = 0, RS_GTU, RS_GTGT, RS_GTG, RS_YY, RS_UY, RS_YGT, RS_PER_POINT };

enum RMode : int {
    R_ACCEPT = 1,  // x ← x + a_acc·u                       optim.jl:136,140
    R_DIR = 2,     // u ← −g + β·u ; Σ g·u, Σ u·u            cg_flavours.jl:10-12, nocedal.jl:56, wolfe.jl:240
    R_TRIAL = 4,   // NPTS × { xp = x + a_j·u ; g⁺ = ∇f(xp) ; all trial sums }   cg_utils.jl:4-23
    R_INIT = 8,    // u = −∇f(x) ; Σ f, Σ g·g                 optim.jl:25-26, cg_flavours.jl:29
    R_RESET = 16,  // u = −∇f(x) ; Σ g·u, Σ u·u               wolfe.jl:129
    R_UPG = 32,    // Σ (u + ∇f(x))²                          wolfe.jl:123
    R_GRAD = 64,   // gout = ∇f(x)                            (materialise for Results.gradient)
    R_GRADT = 128, // gout = ∇f(x + a_0·u)                    (rare: LinearAlgebra.norm scaled path on g⁺)
    R_PROJ = 256   // solvesystem: x2 ← x2 + m·∇f(x + a_0·u) ; g⁺ = ∇f(x2) ; trial sums of g⁺ against ∇f(x), u
                   //                                          solve_system.jl:169-177,199-204,239-253  (m = P.beta)
};

// Launch scalars kept in device memory for launches armed by the on-device controller
// (cgo_ctl.hpp): written by the controller after launch k, read by launch k+1 — no host in between.
struct CtlArgs {
    double a_acc, beta;
    double a[7];   // = MAXP
    long long go;  // 0: the controller stopped — the launch is a no-op
};

#ifndef CGO_RTC
// Device block of the on-device controller: its config and state, the argument block the armed launches read, and the
// number of the round on the DEVICE — record slot and sequence word derive from it, so an armed launch carries no
// per-round host argument (and batches of rounds can replay from one hipGraph).
struct CtlDev { CtlConfig cfg; CtlState st; CtlArgs args; unsigned long long round; };
constexpr int PIPE_RING = 64;   // records in flight
static_assert(sizeof(CtlDev) % 8 == 0 && sizeof(CtlRecord) % 8 == 0, "controller blocks are copied as 8-byte words");
#endif

struct RParams {
    double *x; double *u; double *gout; const double *p0;
    long long n;
    double a_acc, beta;
    double a[MAXP];
    double s0;
    double *partials;
    const CtlArgs *ctl;  // non-null: a_acc, beta, a[] come from device memory instead of the arguments
    double *x2;          // R_PROJ: the second iterate buffer (`x_next` of solve_system.jl:82)
    // Where the updated x and u go: = x, u (in place), or a second pair of buffers the backend swaps in after the
    // launch (ping-pong).  Reading and writing the SAME 0.8-GB arrays in one pure-HBM stream costs ≈ 10 % against
    // writing to other arrays (scripts/tune/rw_mix.hip, n = 1e8: R x,u,D / W x,u in place 712–720 µs, out of place
    // 648–652 µs); at Infinity-Cache sizes the extra footprint costs more than it gains, so only BIG launches use it.
    double *xo; double *uo;
    Tail tail;
};

// Workgroup reduction of N per-lane accumulators → one row of `partials`.
//
// The obvious form — a 6-step __shfl_down tree per accumulator — is 6·N DEPENDENT cross-lane moves per wave (336 for
// the 56-slot rows of a 7-point launch), each a ds_bpermute round trip of ≈ 100 cycles issued one after the other:
// ≈ 17 µs per workgroup, on the critical path of every launch at least once and ≈ 4× that when a CU runs 16
// workgroups back to back (measured: n = 1.25e7, 7 points, 512 / 1024 / 2048 / 4096 workgroups = 94 / 105 / 117 / 152 µs
// against 75–79 µs for the same bytes without a tail; gpurun_out/r02_ab).  Here instead (N a multiple of 8):
//   * three HALVING exchanges (lane ^ 32, ^ 16, ^ 8): a lane sends the half of its slots its partner will keep and adds
//     the half it receives — N/2 + N/4 + N/8 moves, after which lane 8g + r holds N/8 slots (those of group g) summed
//     over the 8 lanes that share r;
//   * three full steps (^ 4, ^ 2, ^ 1) on the remaining N/8 values.
// 7N/8 + 3N/8 = 10N/8 moves instead of 6N (70 for N = 56), and all moves of a step are independent, so their
// latencies overlap.  Fixed pattern ⇒ bit-reproducible; the order of additions differs from the tree's, which only a
// sum's last bits can see.
// ---- fused reduction tail ------------------------------------------------------------------------------------------------
// A launch used to be followed by one or two k_finalize_t launches (rows → sums → pinned host): 2 × 4.2 µs of dependent
// launches on the critical path of EVERY line-search decision — a quarter of an iteration at n = 1e6
// (profiles/r02_gaps_after_c2_n1e6.json).  Instead the workgroup that ARRIVES LAST finishes the job:
//   level 1: workgroups are grouped by 64 (blockIdx / 64); each leaves its row and takes a ticket of its group; the last
//            of a group sums the group's rows into one row of partials2;
//   level 2: group finishers take a ticket of the launch; the last one sums the ≤ 64 group rows, leaves the result in
//            `out`, publishes it to the host block and releases the sequence word.
// WHO is last varies from launch to launch; WHAT is summed in which order does not (groups are fixed by blockIdx, rows
// are read in index order with the interleave G = BLOCK / N of k_finalize_t<N, BLOCK>), so the sums are bit-reproducible
// and equal to those of the finalize launches.
//
// Visibility without fences.  The textbook form — row, __threadfence(), ticket; ticket, __threadfence(), rows — is
// correct and was measured first: on gfx950 an agent-scope fence writes back and invalidates the XCD's whole L2
// (buffer_wbl2 / buffer_inv), once per WORKGROUP here: n = 1e6 launch 14 → 33 µs, n = 1e8 660 → 1040 µs.  So instead
// every slot of the row buffers is a self-validating mailbox:
//   * slots hold TAIL_EMPTY — a SIGNALLING-NaN bit pattern, which no addition can produce — whenever no launch is in
//     flight (filled at context creation; these buffers are used by fused launches only);
//   * a row is written with agent-scope atomic exchanges; the ticket was taken even earlier (it only says "arrived") —
//     no fence, no wait in between;
//   * the finisher TAKES each slot with an agent-scope atomic exchange that puts TAIL_EMPTY back, and asks again while it
//     still sees TAIL_EMPTY (the writer issued its row before it could learn it was not last, so the row arrives; the
//     poll is bounded anyway: it would fall through with the NaN and count an error that cgo_solver_results reports).
// All ≤ 16 exchanges of a lane are in flight before the first is looked at: one memory round trip per level.
// Counters return to zero and slots to TAIL_EMPTY before the launch ends.  The host block validates itself the same way
// (tail_check_term in cgo_kernels.hip.hpp); T.strict = the formal __threadfence_system() + release store instead.
constexpr unsigned long long TAIL_EMPTY = 0xFFF4DEADFFF4DEADull;   // sNaN; both halves equal: hipMemsetD32 fills it
constexpr int TAIL_SPIN = 1 << 22;

// a slot is only ever touched by read-modify-write atomics — the same path through the memory system as the tickets,
// which is the one last-workgroup reductions have always relied on across XCDs
__device__ inline void tail_put(double *slot, double v) {
    (void)__hip_atomic_exchange(reinterpret_cast<unsigned long long *>(slot), (unsigned long long)__double_as_longlong(v),
                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int N>
__device__ inline double tail_sum(double *rows, int nrows, double *fs, unsigned int *err) {
    constexpr int G = BLOCK / N;
    constexpr int L = (TAIL_GROUP + G - 1) / G;
    const int tid = threadIdx.x;
    if (tid < G * N) {
        unsigned long long *q = reinterpret_cast<unsigned long long *>(rows);
        unsigned long long b[L];
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const int i = tid + k * G * N;   // flat index: row i / N, slot i % N — lane (g, s) owns rows g, g + G, …
            b[k] = (i < nrows * N) ? __hip_atomic_exchange(q + i, TAIL_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const int i = tid + k * G * N;
            for (int spin = 0; b[k] == TAIL_EMPTY && spin < TAIL_SPIN; ++spin) {
                __builtin_amdgcn_s_sleep(1);
                b[k] = __hip_atomic_exchange(q + i, TAIL_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (b[k] == TAIL_EMPTY) atomicAdd(err, 1u);   // never seen; the host turns it into an error (tail_errors)
            t += __longlong_as_double((long long)b[k]);   // (+0.0 for absent rows changes nothing: t starts at +0.0)
        }
        fs[tid] = t;
    }
    __syncthreads();
    double r = 0.0;
    if (tid < N) {
#pragma unroll
        for (int g = 0; g < G; ++g) r += fs[g * N + tid];
    }
    __syncthreads();   // fs is used again by the second level
    return r;
}

#ifndef CGO_RTC
// A round's record → its slot of the pinned host ring + the slot's word.  Self-validating like the sums block (word =
// check over the record's 8-byte words and the round number), or formally fenced (T.strict).
__device__ inline void tail_publish_record(const Tail &T, const CtlRecord &sr, unsigned long long round, unsigned long long *terms) {
    constexpr int WR = sizeof(CtlRecord) / 8;
    static_assert(WR <= BLOCK, "one lane per record word");
    const int tid = threadIdx.x;
    unsigned long long *rec_host = reinterpret_cast<unsigned long long *>(reinterpret_cast<CtlRecord *>(T.ctl_rec) + (round % PIPE_RING));
    unsigned long long *seq_host = T.ctl_seq + (round % PIPE_RING);
    const unsigned long long *w = reinterpret_cast<const unsigned long long *>(&sr);
    if (T.strict) {
        if (tid < WR) { rec_host[tid] = w[tid]; __threadfence_system(); }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(seq_host, round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (tid < WR) {
        __hip_atomic_store(rec_host + tid, w[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        terms[tid] = tail_check_term(w[tid], tid);
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long c = tail_check_seq(round + 1);
        for (int i = 0; i < WR; ++i) c += terms[i];
        __hip_atomic_store(seq_host, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The finisher of a controller-armed launch: sums → ctl_step() → state and the NEXT launch's arguments (device memory,
// read by that launch after this one has ended) → the round's record (host).  What k_finalize_ctl does in a launch of
// its own; one lane runs the scalar state machine, the blocks move as 8-byte words, one lane each.
// Multi-rank armed launch: this rank's N sums → the world's sums, between the GPUs themselves (SURVEY.md §8(e) "fast path").
// Lane t < N of wave 0 holds slot t.  The block goes — N values and a check word over (round + 1, values), system-scope
// relaxed stores in any order: self-validating like the host block — into slot [buffer][this rank] of EVERY rank's mailbox,
// over xGMI for the peers; then the four waves poll THIS rank's mailbox, two source ranks each, until every block validates
// for this round, and lanes t < N add them in rank order: bitwise the same sums on every rank.  Two buffers on the round's
// parity: a rank can publish round r + 1 only after it has seen every peer's round r, i.e. after they all finished
// reading round r − 1 — the buffer it now overwrites.  The poll is bounded (a NaN, an error count and a stopped controller).
template <int N>
__device__ inline double tail_exchange(const Tail &T, double v, unsigned long long seq, long long &ticks) {
    __shared__ double xb[8][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long t0 = wall_clock64();
    const int buf = (int)(seq & 1);
    if (wave == 0) {
        unsigned long long c = (lane < N) ? tail_check_term((unsigned long long)__double_as_longlong(v), lane) : 0ull;
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) c += __shfl_xor(c, m, 64);
        c += tail_check_seq(seq);
        for (int r = 0; r < T.xw; ++r) {
            double *dst = T.xmail[r] + ((size_t)buf * T.xw + T.xme) * XSLOT;
            if (lane < N) __hip_atomic_store(dst + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (lane == 0) __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst + 64), c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    for (int r = wave; r < T.xw; r += BLOCK / 64) {
        const double *src = T.xmail[T.xme] + ((size_t)buf * T.xw + r) * XSLOT;
        double val = 0.0;
        bool ok = false;
        for (int spin = 0; !ok && spin < TAIL_SPIN; ++spin) {
            val = (lane < N) ? __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
            unsigned long long c = (lane < N) ? tail_check_term((unsigned long long)__double_as_longlong(val), lane) : 0ull;
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) c += __shfl_xor(c, m, 64);
            const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(src + 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ok = (c + tail_check_seq(seq)) == w;   // (wave-uniform: every lane holds the same c and reads the same word)
            if (!ok) __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) { val = __longlong_as_double((long long)TAIL_EMPTY); if (lane == 0) atomicAdd(T.tickets + TAIL_GROUP + 1, 1u); }
        xb[r][lane] = val;
    }
    __syncthreads();
    double g = 0.0;
    if (tid < N) for (int r = 0; r < T.xw; ++r) g += xb[r][tid];
    ticks = wall_clock64() - t0;
    return g;
}

template <int N>
__device__ inline void tail_ctl(const Tail &T, double v) {
    constexpr int WD = sizeof(CtlDev) / 8, WR = sizeof(CtlRecord) / 8;
    static_assert(WD <= BLOCK, "one lane per word");
    __shared__ double fin[CTL_NSUMS];
    __shared__ CtlDev sd;
    __shared__ CtlRecord sr;
    __shared__ unsigned long long terms[WR];
    const int tid = threadIdx.x;
    CtlDev *d = reinterpret_cast<CtlDev *>(T.ctl);
    if (tid < WD) reinterpret_cast<unsigned long long *>(&sd)[tid] = reinterpret_cast<const unsigned long long *>(d)[tid];
    __syncthreads();
    long long xticks = 0;
    if (T.xw > 1) v = tail_exchange<N>(T, v, T.xseq0 + sd.round + 1, xticks);   // the world's sums instead of this shard's
    if (tid < CTL_NSUMS) { const double f = (tid < N) ? v : 0.0; fin[tid] = f; sr.sums[tid] = f; }
    if (tid < N) T.out[tid] = v;
    __syncthreads();
    const unsigned long long round = sd.round;
    if (tid == 0) {
        ctl_decide(sd.cfg, sd.st, fin, sr);
        sr.xwait = xticks;
        CtlArgs a;
        a.a_acc = sd.st.a_acc; a.beta = sd.st.beta; a.go = sd.st.go;
        for (int j = 0; j < CTL_MAXP; ++j) a.a[j] = sd.st.a[j];
        sd.args = a;
        sd.round = round + 1;
    }
    __syncthreads();
    if (tid < WD) reinterpret_cast<unsigned long long *>(d)[tid] = reinterpret_cast<const unsigned long long *>(&sd)[tid];
    tail_publish_record(T, sr, round, terms);
}

// An armed launch that finds the controller stopped does nothing — except that its round still gets a record (npts = −1)
// and the round counter moves on: workgroup 0 does that.
__device__ inline void tail_ctl_idle(const Tail &T) {
    constexpr int WR = sizeof(CtlRecord) / 8;
    __shared__ CtlRecord sr;
    __shared__ unsigned long long terms[WR];
    __shared__ unsigned long long round_s;
    const int tid = threadIdx.x;
    CtlDev *d = reinterpret_cast<CtlDev *>(T.ctl);
    if (tid == 0) {
        round_s = d->round;
        for (int i = 0; i < CTL_NSUMS; ++i) sr.sums[i] = 0.0;
        sr.a_acc = 0.0; sr.beta = 0.0;
        for (int j = 0; j < CTL_MAXP; ++j) sr.a[j] = 0.0;
        sr.npts = -1; sr.accepted = 0; sr.xwait = 0;
        d->round = round_s + 1;
    }
    __syncthreads();
    tail_publish_record(T, sr, round_s, terms);
}
#endif

// the finished sums (lane t < N holds slot t) → device copy and host block
template <int N>
__device__ inline void tail_publish(const Tail &T, double v) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        if (tid < N) T.out[tid] = v;
        if (T.host_out && T.strict) {
            if (tid < N) { T.host_out[tid] = v; __threadfence_system(); }
            if (tid == 0) __hip_atomic_store(T.host_seq, T.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (T.host_out) {   // self-validating block: values and check word in any order (tail_check_term)
            unsigned long long c = 0ull;
            if (tid < N) {
                __hip_atomic_store(T.host_out + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                c = tail_check_term((unsigned long long)__double_as_longlong(v), tid);
            }
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) c += __shfl_xor(c, m, 64);
            if (tid == 0) __hip_atomic_store(T.host_seq, c + tail_check_seq(T.seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// `ticket1`: lane 0's ticket of its group, taken BEFORE the workgroup's own reduction (store_partials_n) so that its
// round trip hides behind the cross-lane moves — a ticket says "arrived at the tail", not "row written"; the slots
// say the rest.  The group finisher likewise takes the launch ticket while its group's rows are in flight.
template <int N, bool CTL>
__device__ inline void finish_tail(const Tail &T, double *partials, double own, unsigned ticket1) {
    static_assert(N <= 64, "the row is written by lanes of wave 0");
    __shared__ double fs[(BLOCK / N) * N];
    __shared__ int last;
    const int tid = threadIdx.x;
    const int nb = gridDim.x, grp = blockIdx.x / TAIL_GROUP, ngroups = (nb + TAIL_GROUP - 1) / TAIL_GROUP;
    const int left = nb - grp * TAIL_GROUP, in_group = left < TAIL_GROUP ? left : TAIL_GROUP;
    if (tid < N) tail_put(partials + (size_t)blockIdx.x * N + tid, own);
    if (tid == 0) last = (ticket1 == (unsigned)(in_group - 1)) ? 1 : 0;
    __syncthreads();
    if (!last) return;
    unsigned ticket2 = 0;
    if (ngroups > 1 && tid == 0) ticket2 = atomicAdd(&T.tickets[TAIL_GROUP], 1u);
    double v = tail_sum<N>(partials + (size_t)grp * TAIL_GROUP * N, in_group, fs, T.tickets + TAIL_GROUP + 1);
    if (tid == 0) __hip_atomic_store(&T.tickets[grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ngroups > 1) {
        if (tid < N) tail_put(T.partials2 + (size_t)grp * N + tid, v);
        if (tid == 0) last = (ticket2 == (unsigned)(ngroups - 1)) ? 1 : 0;
        __syncthreads();
        if (!last) return;
        v = tail_sum<N>(T.partials2, ngroups, fs, T.tickets + TAIL_GROUP + 1);
        if (tid == 0) __hip_atomic_store(&T.tickets[TAIL_GROUP], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifndef CGO_RTC
    if (CTL && T.ctl) { tail_ctl<N>(T, v); return; }   // (grid-stride accept+dir+trial launches only: CTL)
#endif
    tail_publish<N>(T, v);
}

// Rows that a launch did not sum itself (≥ 1024 workgroups; the stored-gradient, L-BFGS and log-sum-exp launches) in ONE
// launch instead of two: workgroup g sums rows [64g, 64g + 64) — plain loads, the producing launch has ended; the order
// of k_finalize_t<N, BLOCK> — leaves the sum in its slot row and holds a ticket taken on arrival; the last one sums the
// ≤ 64 slot rows and publishes (same mailbox slots, same self-validating host block as finish_tail).
template <int N>
__global__ __launch_bounds__(BLOCK) void k_finalize_one(const double *partials, int rows, const Tail T) {
    static_assert(N <= 64, "one wave publishes the block");
    constexpr int G = BLOCK / N, L = (TAIL_GROUP + G - 1) / G;
    __shared__ double fs[G * N];
    __shared__ int last;
    const int tid = threadIdx.x, grp = blockIdx.x, ngroups = gridDim.x;
    unsigned ticket = 0;
    if (ngroups > 1 && tid == 0) ticket = atomicAdd(&T.tickets[TAIL_GROUP], 1u);
    const int left = rows - grp * TAIL_GROUP, in_group = left < TAIL_GROUP ? left : TAIL_GROUP;
    const double *p = partials + (size_t)grp * TAIL_GROUP * N;
    if (tid < G * N) {
        double v[L];
#pragma unroll
        for (int k = 0; k < L; ++k) { const int i = tid + k * G * N; v[k] = (i < in_group * N) ? p[i] : 0.0; }
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < L; ++k) t += v[k];
        fs[tid] = t;
    }
    __syncthreads();
    double v = 0.0;
    if (tid < N) {
#pragma unroll
        for (int g = 0; g < G; ++g) v += fs[g * N + tid];
    }
    __syncthreads();
    if (ngroups > 1) {
        if (tid < N) tail_put(T.partials2 + (size_t)grp * N + tid, v);
        if (tid == 0) last = (ticket == (unsigned)(ngroups - 1)) ? 1 : 0;
        __syncthreads();
        if (!last) return;
        v = tail_sum<N>(T.partials2, ngroups, fs, T.tickets + TAIL_GROUP + 1);
        if (tid == 0) __hip_atomic_store(&T.tickets[TAIL_GROUP], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    tail_publish<N>(T, v);
}

// Cross-lane moves WITHOUT the LDS crossbar.  __shfl_xor lowers to ds_bpermute_b32 — a round trip through the LDS hardware of
// ≈ 100+ cycles per 32-bit half — and the transpose-reduce below needs 10N/8 of them per wave: 1.1–1.5 µs per workgroup for
// a 24-slot row, measured inside the resident kernel (round 3).  gfx950 moves the same data inside the VALU:
//   lane ^ 32, ^ 16 : v_permlane32_swap / v_permlane16_swap exchange the upper (odd-row) half of one register with the
//                     lower (even-row) half of another — exactly the "send the half the partner keeps" step, for a PAIR of
//                     slots at once and without the select;
//   lane ^ 8        : v_mov_b32_dpp row_ror:8;   the last three levels: row_half_mirror, quad_perm [1,0,3,2], [2,3,0,1].
// (The pairing of the last three levels is (i, 7−i), (i, i^1), (i, i^2) instead of (i^4, i^2, i^1): another fixed order of the
//  same additions.)
typedef unsigned int cgo_u2 __attribute__((ext_vector_type(2)));
// (x, y) → x' = x + partner's x on the lanes with bit 5 (ROW32) / bit 4 (!ROW32) clear, y + partner's y on the others
template <bool ROW32>
__device__ inline double swap_add(double x, double y) {
    const unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x);
    const unsigned yl = (unsigned)__double2loint(y), yh = (unsigned)__double2hiint(y);
    cgo_u2 lo, hi;
    if (ROW32) { lo = __builtin_amdgcn_permlane32_swap(xl, yl, false, false); hi = __builtin_amdgcn_permlane32_swap(xh, yh, false, false); }
    else { lo = __builtin_amdgcn_permlane16_swap(xl, yl, false, false); hi = __builtin_amdgcn_permlane16_swap(xh, yh, false, false); }
    // lower half: (own x, partner's x); upper half: (partner's y, own y)
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
template <int CTRL>
__device__ inline double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_ROW_ROR8 = 0x128, DPP_HALF_MIRROR = 0x141, DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E;

// The workgroup's N sums: lane t < N returns slot t (the other lanes 0.0).  Transpose-reduce, see above.
template <int N>
__device__ inline double wg_reduce_n(double (&acc)[N]) {
    __shared__ double sm[BLOCK / 64][N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CGO_TREE_TAIL   // A/B: the slot-major __shfl_down tree of round 1
    if (false) {
#else
    if (N % 8 == 0) {
#endif
        constexpr int H0 = N / 2, H1 = N / 4, H2 = N / 8;
#ifdef CGO_BPERMUTE_TAIL   // A/B: the same transpose-reduce on ds_bpermute (round 2)
        {   // ^32: lanes 0–31 keep slots [0, H0), lanes 32–63 keep [H0, N)
            const bool up = (lane & 32) != 0;
#pragma unroll
            for (int c = 0; c < H0; c += H2) {
#pragma unroll
                for (int j = c; j < c + H2; ++j) {
                    const double send = up ? acc[j] : acc[j + H0];
                    const double keep = up ? acc[j + H0] : acc[j];
                    acc[j] = keep + __shfl_xor(send, 32, 64);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {   // ^16 on the H0 kept values
            const bool up = (lane & 16) != 0;
#pragma unroll
            for (int c = 0; c < H1; c += H2) {
#pragma unroll
                for (int j = c; j < c + H2; ++j) {
                    const double send = up ? acc[j] : acc[j + H1];
                    const double keep = up ? acc[j + H1] : acc[j];
                    acc[j] = keep + __shfl_xor(send, 16, 64);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {   // ^8 on the H1 kept values
            const bool up = (lane & 8) != 0;
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                const double send = up ? acc[j] : acc[j + H2];
                const double keep = up ? acc[j + H2] : acc[j];
                acc[j] = keep + __shfl_xor(send, 8, 64);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 4; m > 0; m >>= 1) {
#pragma unroll
            for (int j = 0; j < H2; ++j) acc[j] += __shfl_xor(acc[j], m, 64);
        }
#else
        // The exchanges run in chunks of H2 slots with a scheduling fence between chunks: unfenced, the compiler hoists
        // all moves of a step and their temporaries push the 7-point kernel's register count up a class — three waves
        // per SIMD instead of four, which cost the pure-HBM launches 12 % (n = 1e8: 681 → 767 µs; round 2).
        // ^32: lanes 0–31 keep slots [0, H0), lanes 32–63 keep [H0, N)
#pragma unroll
        for (int c = 0; c < H0; c += H2) {
#pragma unroll
            for (int j = c; j < c + H2; ++j) acc[j] = swap_add<true>(acc[j], acc[j + H0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ^16 on the H0 kept values
#pragma unroll
        for (int c = 0; c < H1; c += H2) {
#pragma unroll
            for (int j = c; j < c + H2; ++j) acc[j] = swap_add<false>(acc[j], acc[j + H1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        {   // ^8 on the H1 kept values
            const bool up = (lane & 8) != 0;
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                const double send = up ? acc[j] : acc[j + H2];
                const double keep = up ? acc[j + H2] : acc[j];
                acc[j] = keep + dpp_mov<DPP_ROW_ROR8>(send);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < H2; ++j) acc[j] += dpp_mov<DPP_HALF_MIRROR>(acc[j]);
#pragma unroll
        for (int j = 0; j < H2; ++j) acc[j] += dpp_mov<DPP_QUAD_XOR1>(acc[j]);
#pragma unroll
        for (int j = 0; j < H2; ++j) acc[j] += dpp_mov<DPP_QUAD_XOR2>(acc[j]);
#endif
        if ((lane & 7) == 0) {   // lane 8g holds the H2 slots of group g = 4·b5 + 2·b4 + b3
            const int g = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
#pragma unroll
            for (int j = 0; j < H2; ++j) sm[wave][g * H2 + j] = acc[j];
        }
    } else {   // step-major tree: the N moves of a step are independent
#ifdef CGO_TREE_TAIL
#pragma unroll
        for (int s = 0; s < N; ++s) acc[s] = wave_sum(acc[s]);
#else
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int s = 0; s < N; ++s) acc[s] += __shfl_down(acc[s], off, 64);
        }
#endif
        if (lane == 0) {
#pragma unroll
            for (int s = 0; s < N; ++s) sm[wave][s] = acc[s];
        }
    }
    __syncthreads();
    const double own = (tid < N) ? (sm[0][tid] + sm[1][tid]) + (sm[2][tid] + sm[3][tid]) : 0.0;
    __syncthreads();   // sm may be written again by the caller's next reduction (resident passes)
    return own;
}

template <int N, bool CTL = false>
__device__ inline void store_partials_n(double (&acc)[N], double *partials, const Tail &T) {
    const int tid = threadIdx.x;
    unsigned ticket1 = 0;
    if (T.tickets && tid == 0) ticket1 = atomicAdd(&T.tickets[blockIdx.x / TAIL_GROUP], 1u);   // in flight during the reduction below
    const double own = wg_reduce_n<N>(acc);
    if (T.tickets) finish_tail<N, CTL>(T, partials, own, ticket1);
    else if (tid < N) partials[(size_t)blockIdx.x * N + tid] = own;
}

template <int NPTS> struct RW { static constexpr int W = (NPTS == 1) ? NR1 : (NPTS == 3 ? NR : (NPTS == 5 ? NR5 : NR7)); static constexpr int GU = RS_PER_POINT * NPTS, UU = GU + 1; };

template <class Obj, int MODE, int NPTS>
__device__ inline void cg_pair(const RParams &P, d2 &x, d2 &u, d2 p, double (&acc)[RW<NPTS>::W], bool &wx, bool &wu, d2 &gout) {
    constexpr int R_GU = RW<NPTS>::GU, R_UU = RW<NPTS>::UU;
    if (MODE & R_ACCEPT) {
        x.x = x.x + P.a_acc * u.x;
        x.y = x.y + P.a_acc * u.y;
        wx = true;
    }
    d2 g;
    double f0 = 0.0;
    constexpr bool need_g = (MODE & (R_DIR | R_TRIAL | R_INIT | R_RESET | R_UPG | R_GRAD | R_PROJ)) != 0;
    g = d2{0.0, 0.0};
    if (need_g) Obj::eval2(x, p, P.s0, f0, g);  // g = ∇f(x), recomputed — never read from HBM
    if (MODE & R_INIT) {
        acc[RS_F] += f0;
        acc[RS_GTGT] = dsum(acc[RS_GTGT], g.x, g.x);
        acc[RS_GTGT] = dsum(acc[RS_GTGT], g.y, g.y);
        u.x = -g.x; u.y = -g.y;
        wu = true;
    }
    if (MODE & (R_DIR | R_RESET)) {
        d2 un;
        if (MODE & R_DIR) { un.x = -g.x + P.beta * u.x; un.y = -g.y + P.beta * u.y; }
        else { un.x = -g.x; un.y = -g.y; }
        acc[R_GU] = dsum(acc[R_GU], g.x, un.x); acc[R_GU] = dsum(acc[R_GU], g.y, un.y);
        acc[R_UU] = dsum(acc[R_UU], un.x, un.x); acc[R_UU] = dsum(acc[R_UU], un.y, un.y);
        u = un;
        wu = true;
    }
    if (MODE & R_UPG) {
        const double t0 = u.x + g.x, t1 = u.y + g.y;
        acc[R_UU] = dsum(acc[R_UU], t0, t0); acc[R_UU] = dsum(acc[R_UU], t1, t1);
    }
    if (MODE & R_GRAD) gout = g;
    if (MODE & R_GRADT) {
        d2 xp;
        double fd = 0.0;
        xp.x = x.x + P.a[0] * u.x;
        xp.y = x.y + P.a[0] * u.y;
        Obj::eval2(xp, p, P.s0, fd, gout);
    }
    if (MODE & R_PROJ) {  // gout carries x2 in and out (the caller loads/stores it)
        d2 z, gz, gt;
        double fz = 0.0;
        z.x = x.x + P.a[0] * u.x;
        z.y = x.y + P.a[0] * u.y;
        Obj::eval2(z, p, P.s0, fz, gz);                 // df_xp of the accepted trial (solve_system.jl:46)
        gout.x = gout.x + P.beta * gz.x;                // x_next[i] = x_next[i] + m*df_xp[i]   (:250-252)
        gout.y = gout.y + P.beta * gz.y;
        Obj::eval2(gout, p, P.s0, acc[RS_F], gt);       // f_x_next = fdf!(df_xp, x_next)        (:177)
        const double y0 = gt.x - g.x, y1 = gt.y - g.y;  // getβ(β_config, df_xp, df_x, u)       (:199-204)
        acc[RS_GTU] = dsum(acc[RS_GTU], gt.x, u.x);   acc[RS_GTU] = dsum(acc[RS_GTU], gt.y, u.y);
        acc[RS_GTGT] = dsum(acc[RS_GTGT], gt.x, gt.x); acc[RS_GTGT] = dsum(acc[RS_GTGT], gt.y, gt.y);
        acc[RS_GTG] = dsum(acc[RS_GTG], gt.x, g.x);   acc[RS_GTG] = dsum(acc[RS_GTG], gt.y, g.y);
        acc[RS_YY] = dsum(acc[RS_YY], y0, y0);       acc[RS_YY] = dsum(acc[RS_YY], y1, y1);
        acc[RS_UY] = dsum(acc[RS_UY], u.x, y0);      acc[RS_UY] = dsum(acc[RS_UY], u.y, y1);
        acc[RS_YGT] = dsum(acc[RS_YGT], y0, gt.x);    acc[RS_YGT] = dsum(acc[RS_YGT], y1, gt.y);
    }
    if (MODE & R_TRIAL) {
#pragma unroll
        for (int j = 0; j < NPTS; ++j) {
            const int b = RS_PER_POINT * j;
            d2 xp, gt;
            xp.x = x.x + P.a[j] * u.x;
            xp.y = x.y + P.a[j] * u.y;
            Obj::eval2(xp, p, P.s0, acc[b + RS_F], gt);
            const double y0 = gt.x - g.x, y1 = gt.y - g.y;
            acc[b + RS_GTU] = dsum(acc[b + RS_GTU], gt.x, u.x);   acc[b + RS_GTU] = dsum(acc[b + RS_GTU], gt.y, u.y);
            acc[b + RS_GTGT] = dsum(acc[b + RS_GTGT], gt.x, gt.x); acc[b + RS_GTGT] = dsum(acc[b + RS_GTGT], gt.y, gt.y);
            acc[b + RS_GTG] = dsum(acc[b + RS_GTG], gt.x, g.x);   acc[b + RS_GTG] = dsum(acc[b + RS_GTG], gt.y, g.y);
            acc[b + RS_YY] = dsum(acc[b + RS_YY], y0, y0);       acc[b + RS_YY] = dsum(acc[b + RS_YY], y1, y1);
            acc[b + RS_UY] = dsum(acc[b + RS_UY], u.x, y0);      acc[b + RS_UY] = dsum(acc[b + RS_UY], u.y, y1);
            acc[b + RS_YGT] = dsum(acc[b + RS_YGT], y0, gt.x);    acc[b + RS_YGT] = dsum(acc[b + RS_YGT], y1, gt.y);
        }
    }
}

// odd tail element (objectives that are not pair-only)
template <class Obj, int MODE, int NPTS>
__device__ inline void cg_single(const RParams &P, long long i, double (&acc)[RW<NPTS>::W]) {
    constexpr int R_GU = RW<NPTS>::GU, R_UU = RW<NPTS>::UU;
    double x = P.x[i];
    double u = (MODE & (R_ACCEPT | R_DIR | R_TRIAL | R_UPG | R_GRADT | R_PROJ)) ? P.u[i] : 0.0;
    const double p = Obj::kParam ? P.p0[i] : 0.0;
    if (MODE & R_ACCEPT) { x = x + P.a_acc * u; P.xo[i] = x; }
    double g = 0.0, f0 = 0.0;
    Obj::eval1(x, p, P.s0, f0, g);
    if (MODE & R_INIT) { acc[RS_F] += f0; acc[RS_GTGT] = dsum(acc[RS_GTGT], g, g); P.uo[i] = -g; }
    if (MODE & (R_DIR | R_RESET)) {
        const double un = (MODE & R_DIR) ? (-g + P.beta * u) : -g;
        acc[R_GU] = dsum(acc[R_GU], g, un); acc[R_UU] = dsum(acc[R_UU], un, un);
        P.uo[i] = un; u = un;
    }
    if (MODE & R_UPG) { const double t = u + g; acc[R_UU] = dsum(acc[R_UU], t, t); }
    if (MODE & R_GRAD) P.gout[i] = g;
    if (MODE & R_GRADT) { double fd = 0.0, gg; Obj::eval1(x + P.a[0] * u, p, P.s0, fd, gg); P.gout[i] = gg; }
    if (MODE & R_PROJ) {
        double fz = 0.0, gz, gt;
        Obj::eval1(x + P.a[0] * u, p, P.s0, fz, gz);
        const double xn = P.x2[i] + P.beta * gz;
        P.x2[i] = xn;
        Obj::eval1(xn, p, P.s0, acc[RS_F], gt);
        const double y = gt - g;
        acc[RS_GTU] = dsum(acc[RS_GTU], gt, u); acc[RS_GTGT] = dsum(acc[RS_GTGT], gt, gt); acc[RS_GTG] = dsum(acc[RS_GTG], gt, g);
        acc[RS_YY] = dsum(acc[RS_YY], y, y); acc[RS_UY] = dsum(acc[RS_UY], u, y); acc[RS_YGT] = dsum(acc[RS_YGT], y, gt);
    }
    if (MODE & R_TRIAL) {
#pragma unroll
        for (int j = 0; j < NPTS; ++j) {
            const int b = RS_PER_POINT * j;
            const double xp = x + P.a[j] * u;
            double gt;
            Obj::eval1(xp, p, P.s0, acc[b + RS_F], gt);
            const double y = gt - g;
            acc[b + RS_GTU] = dsum(acc[b + RS_GTU], gt, u); acc[b + RS_GTGT] = dsum(acc[b + RS_GTGT], gt, gt); acc[b + RS_GTG] = dsum(acc[b + RS_GTG], gt, g);
            acc[b + RS_YY] = dsum(acc[b + RS_YY], y, y); acc[b + RS_UY] = dsum(acc[b + RS_UY], u, y); acc[b + RS_YGT] = dsum(acc[b + RS_YGT], y, gt);
        }
    }
}

#if defined(CGO_STAMPS) && !defined(CGO_RTC)
// Diagnostic build only (make EXTRA=-DCGO_STAMPS): per workgroup of the LAST k_cg launch, 100-MHz wall-clock stamps at entry,
// after the streaming loop, after the reduction tail, and the hardware id (XCC in the top half) — read by cgo_debug_stamps().
static __device__ unsigned long long cgo_stamps[4096 * 4];
#endif

// CTL: the controller's code (tail_ctl) is compiled in — k_cg_armed only, so that every other launch stays free of its
// LDS and scratch.
template <class Obj, int MODE, int NPTS, bool BIG, bool CTL>
__device__ inline void cg_launch(const RParams &Pin) {
    constexpr int W = RW<NPTS>::W;
    RParams P = Pin;
#if defined(CGO_STAMPS) && !defined(CGO_RTC)
    const unsigned long long st0 = (unsigned long long)wall_clock64();
#endif
    if (MODE == (R_ACCEPT | R_DIR | R_TRIAL) && P.ctl) {  // wave-uniform scalar loads
        const CtlArgs c = *P.ctl;
        if (!c.go) {
#ifndef CGO_RTC
            if (CTL && P.tail.ctl && blockIdx.x == 0) tail_ctl_idle(P.tail);
#endif
            return;
        }
        P.a_acc = c.a_acc; P.beta = c.beta;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) P.a[j] = c.a[j];
    }
    double acc[W];
#pragma unroll
    for (int s = 0; s < W; ++s) acc[s] = 0.0;
    constexpr bool rd_u = (MODE & (R_ACCEPT | R_DIR | R_TRIAL | R_UPG | R_GRADT | R_PROJ)) != 0;
    constexpr bool wr_g = (MODE & (R_GRAD | R_GRADT)) != 0;
    constexpr bool proj = (MODE & R_PROJ) != 0;  // the gout argument of cg_pair carries x2
    const long long n2 = P.n >> 1;
    long long i, hi, step;
    if (BIG) {
        const long long per = big_chunk_pairs(n2, gridDim.x);
        i = per * blockIdx.x + threadIdx.x;
        hi = (per * blockIdx.x + per < n2) ? per * blockIdx.x + per : n2;
        step = BLOCK;
    } else {
        i = (long long)blockIdx.x * BLOCK + threadIdx.x;
        hi = n2;
        step = (long long)gridDim.x * BLOCK;
    }
    // (Two workgroups per CU put two waves on every SIMD and the issue arbiter serves the OLDER wave first: at n = 1.25e7 with
    // seven trial points the workgroup that reached a CU first runs its loop in 58 µs, the second in 83 µs, on 256 of 256 CUs.
    // Taking turns at priority, an uneven static split, a software prefetch, a max-ILP schedule and a third wave per SIMD were
    // all measured in round 4 and none shortens the launch — a CU delivers a fixed amount of this work per µs: DESIGN.md §7.)
    // two independent 16-B groups per lane per trip (≥ 6 loads in flight)
    for (; i + step < hi; i += 2 * step) {
        d2 xa = ldg2<BIG>(P.x, i), xb = ldg2<BIG>(P.x, i + step);
        d2 ua = rd_u ? ldg2<BIG>(P.u, i) : d2{0.0, 0.0}, ub = rd_u ? ldg2<BIG>(P.u, i + step) : d2{0.0, 0.0};
        const d2 pa = Obj::kParam ? ldg2<BIG>(P.p0, i) : d2{0.0, 0.0};
        const d2 pb = Obj::kParam ? ldg2<BIG>(P.p0, i + step) : d2{0.0, 0.0};
        bool wxa = false, wua = false, wxb = false, wub = false;
        d2 ga, gb;
        if (proj) { ga = ldg2<BIG>(P.x2, i); gb = ldg2<BIG>(P.x2, i + step); }
        cg_pair<Obj, MODE, NPTS>(P, xa, ua, pa, acc, wxa, wua, ga);
        cg_pair<Obj, MODE, NPTS>(P, xb, ub, pb, acc, wxb, wub, gb);
        if (wxa) stg2<BIG>(P.xo, i, xa);
        if (wua) stg2<BIG>(P.uo, i, ua);
        if (wr_g) stg2<BIG>(P.gout, i, ga);
        if (proj) stg2<BIG>(P.x2, i, ga);
        if (wxb) stg2<BIG>(P.xo, i + step, xb);
        if (wub) stg2<BIG>(P.uo, i + step, ub);
        if (wr_g) stg2<BIG>(P.gout, i + step, gb);
        if (proj) stg2<BIG>(P.x2, i + step, gb);
    }
    if (i < hi) {
        d2 xa = ldg2<BIG>(P.x, i);
        d2 ua = rd_u ? ldg2<BIG>(P.u, i) : d2{0.0, 0.0};
        const d2 pa = Obj::kParam ? ldg2<BIG>(P.p0, i) : d2{0.0, 0.0};
        bool wxa = false, wua = false;
        d2 ga;
        if (proj) ga = ldg2<BIG>(P.x2, i);
        cg_pair<Obj, MODE, NPTS>(P, xa, ua, pa, acc, wxa, wua, ga);
        if (wxa) stg2<BIG>(P.xo, i, xa);
        if (wua) stg2<BIG>(P.uo, i, ua);
        if (wr_g) stg2<BIG>(P.gout, i, ga);
        if (proj) stg2<BIG>(P.x2, i, ga);
    }
    if ((P.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) cg_single<Obj, MODE, NPTS>(P, P.n - 1, acc);
    if (MODE == R_ACCEPT || MODE == R_GRAD || MODE == R_GRADT) return;  // no sums
#if defined(CGO_STAMPS) && !defined(CGO_RTC)
    const unsigned long long st1 = (unsigned long long)wall_clock64();
#endif
    // (Pin.tail, not P.tail: the armed finisher indexes the tail's mailbox table with a run-time rank, and a run-time index into
    // the modifiable COPY of the arguments puts the whole copy into scratch memory — every lane then re-reads pointers and
    // steps from there inside the streaming loop: the armed seven-point launch at n = 1.25e7 took 190 µs instead of 85, found
    // in round 4.  cgo_ctl.hpp avoids run-time indices into local arrays on the controller's path for the same reason.)
    store_partials_n<W, CTL>(acc, P.partials, Pin.tail);
#if defined(CGO_STAMPS) && !defined(CGO_RTC)
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        unsigned long long *o = cgo_stamps + 4 * blockIdx.x;
        o[0] = st0; o[1] = st1; o[2] = (unsigned long long)wall_clock64();
        o[3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg(63492);
    }
#endif
}

template <class Obj, int MODE, int NPTS, bool BIG>
__global__ __launch_bounds__(BLOCK) void k_cg(const RParams P) { cg_launch<Obj, MODE, NPTS, BIG, false>(P); }

#ifndef CGO_RTC
// A controller-armed round as ONE launch: grid-stride accept + dir + trial whose finisher runs the controller (tail_ctl).
template <class Obj, int NPTS>
__global__ __launch_bounds__(BLOCK) void k_cg_armed(const RParams P) { cg_launch<Obj, R_ACCEPT | R_DIR | R_TRIAL, NPTS, false, true>(P); }
#endif

}  // namespace dev
}  // namespace cgo
