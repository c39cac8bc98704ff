// cgo_kernels_lse.hip.hpp — two-phase kernels for the log-sum-exp objective
//   f(x) = log Σ_i exp(x_i) + ½λ‖x‖²        (BASELINE config 4; not element-wise)
//   ∇f_i = exp(x_i − lse) + λ x_i
//
// The reference would evaluate this through an arbitrary `fdf!` closure that materialises the
// whole gradient on every line-search trial (src/cg_utils.jl:19).  Here a trial needs only
// ϕ(a) and dϕ(a), and both are reductions over xp = x + a·u:
//     ϕ  = M + log S + ½λ Q          M = max xp,  S = Σ e^{xp−M},  Q = Σ xp²
//     dϕ = T/S + λ R                 T = Σ e^{xp−M} u,  R = Σ xp·u
// so phase 1 (k_lse_stats) is a 16 B/elt read-only pass with an online (max, Σ) merge, and the
// gradient is written once per ACCEPTED step by phase 2 (k_lse_grad, 24–32 B/elt) together with
// the getβ / L-BFGS partial sums.  Shards merge (M, S, T) across ranks in rank order.
#pragma once

#include "cgo_kernels.hip.hpp"

namespace cgo {
namespace dev {

enum LseSlot : int { L_M = 0, L_S = 1, L_T = 2, L_Q = 3, L_R = 4 };  // S_GU = 7, S_UU = 8 as usual
enum LseMode : int { LM_ACCEPT = 1, LM_DIR = 2, LM_NOU = 4 };

struct LseAcc { double m, S, T; };

__device__ inline void lse_push(LseAcc &a, double v, double u) {
    if (v > a.m) {
        const double sc = exp(a.m - v);  // a.m = −inf on the first element → 0
        a.S = a.S * sc + 1.0;
        a.T = a.T * sc + u;
        a.m = v;
    } else {
        const double e = exp(v - a.m);   // NaN input propagates into S, T
        a.S += e;
        a.T += e * u;
    }
}

__device__ __host__ inline void lse_merge(double &m, double &S, double &T, double m2, double S2, double T2) {
    const double M = (m2 > m) ? m2 : m;
    const double s1 = (m == M) ? 1.0 : exp(m - M);
    const double s2 = (m2 == M) ? 1.0 : exp(m2 - M);
    S = S * s1 + S2 * s2;
    T = T * s1 + T2 * s2;
    m = M;
}

__device__ inline void wave_lse(double &m, double &S, double &T) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double m2 = __shfl_down(m, off, 64), S2 = __shfl_down(S, off, 64), T2 = __shfl_down(T, off, 64);
        lse_merge(m, S, T, m2, S2, T2);
    }
}

struct LseParams {
    double *x; double *u; const double *g; double *gt;
    long long n;
    double a_acc, beta, a_trial, lambda;
    double M, S;        // global max / Σ of the trial point (phase 2)
    double *partials;
};

// rows: [M, S, T, Q, R, -, -, gu, uu, -]
__device__ inline void store_partials_lse(LseAcc &la, double (&acc)[NS], double *partials) {
    __shared__ double sm[BLOCK / 64][NS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    wave_lse(la.m, la.S, la.T);
#pragma unroll
    for (int s = 3; s < NS; ++s) acc[s] = wave_sum(acc[s]);
    if (lane == 0) {
        sm[wave][L_M] = la.m; sm[wave][L_S] = la.S; sm[wave][L_T] = la.T;
#pragma unroll
        for (int s = 3; s < NS; ++s) sm[wave][s] = acc[s];
    }
    __syncthreads();
    if (tid == 0) {
        double m = sm[0][L_M], S = sm[0][L_S], T = sm[0][L_T];
        for (int w = 1; w < BLOCK / 64; ++w) lse_merge(m, S, T, sm[w][L_M], sm[w][L_S], sm[w][L_T]);
        double *row = partials + (size_t)blockIdx.x * NS;
        row[L_M] = m; row[L_S] = S; row[L_T] = T;
#pragma unroll
        for (int s = 3; s < NS; ++s) row[s] = (sm[0][s] + sm[1][s]) + (sm[2][s] + sm[3][s]);
    }
}

// Two stages when there are many rows (a 4 096-workgroup launch leaves 320 KB of rows; ONE workgroup streams them at ≈ 25 GB/s:
// 12 µs, round 3 profile of config 4): workgroup b of the first stage merges rows [b·rows_per_block, …) into row b of
// `out_all` (host_out = nullptr), a single workgroup then merges those and publishes.  One stage = gridDim.x == 1.
static __global__ __launch_bounds__(BLOCK) void k_finalize_lse(const double *partials_all, int rows_per_block, int rows_total, double *out_all,
                                                        double *host_out, unsigned long long *host_seq,
                                                        unsigned long long seq) {
    __shared__ double sm[BLOCK / 64][NS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long first = (long long)blockIdx.x * rows_per_block;
    int rows = (int)((long long)rows_total - first < rows_per_block ? (long long)rows_total - first : rows_per_block);
    if (rows < 0) rows = 0;
    const double *partials = partials_all + first * NS;
    double *out = out_all + (size_t)blockIdx.x * NS;
    double m = -INFINITY, S = 0.0, T = 0.0;
    double tot[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) tot[s] = 0.0;
    for (int b = tid; b < rows; b += BLOCK) {
        const double *row = partials + (size_t)b * NS;
        lse_merge(m, S, T, row[L_M], row[L_S], row[L_T]);
#pragma unroll
        for (int s = 3; s < NS; ++s) tot[s] += row[s];
    }
    wave_lse(m, S, T);
#pragma unroll
    for (int s = 3; s < NS; ++s) tot[s] = wave_sum(tot[s]);
    if (lane == 0) {
        sm[wave][L_M] = m; sm[wave][L_S] = S; sm[wave][L_T] = T;
#pragma unroll
        for (int s = 3; s < NS; ++s) sm[wave][s] = tot[s];
    }
    __syncthreads();
    if (tid == 0) {
        m = sm[0][L_M]; S = sm[0][L_S]; T = sm[0][L_T];
        for (int w = 1; w < BLOCK / 64; ++w) lse_merge(m, S, T, sm[w][L_M], sm[w][L_S], sm[w][L_T]);
        double v[NS];
        v[L_M] = m; v[L_S] = S; v[L_T] = T;
#pragma unroll
        for (int s = 3; s < NS; ++s) v[s] = (sm[0][s] + sm[1][s]) + (sm[2][s] + sm[3][s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            out[s] = v[s];
            if (host_out) host_out[s] = v[s];
        }
        if (host_out) {
            __threadfence_system();
            __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// phase 1: statistics of the trial point, optionally fused with accept + direction update.
// REF (round 3): the running maximum — a data-dependent branch with two `exp` on one side, per element, which kept this
// 16 B/element read-only pass at half the HBM rate — is replaced by a FIXED reference P.M = lse of the last evaluated point
// on this line: e_i = exp(xp_i − P.M), S' = Σ e_i, T' = Σ e_i·u_i are plain sums (row slots L_S, L_T; L_M unused) and
// ϕ = P.M + log S' + ½λQ.  The host accepts them while S' is finite and positive (else: this kernel's running-maximum form).
template <int MODE, bool BIG, bool REF>
__global__ __launch_bounds__(BLOCK) void k_lse_stats(const LseParams P) {
    LseAcc la{-INFINITY, 0.0, 0.0};
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.0;
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
    auto one = [&](double &x, double &u, double g, bool store, double *px, double *pu) {
        if (MODE & LM_ACCEPT) { x = x + P.a_acc * u; if (store) *px = x; }
        if (MODE & LM_DIR) {
            const double un = -g + P.beta * u;
            acc[S_GU] += g * un;
            acc[S_UU] += un * un;
            u = un;
            if (store) *pu = un;
        }
        const double xp = (MODE & LM_NOU) ? x : (x + P.a_trial * u);
        const double uu = (MODE & LM_NOU) ? 0.0 : u;
        if (REF) {
            const double e = exp(xp - P.M);
            acc[L_S] += e;
            acc[L_T] = dsum(acc[L_T], e, uu);
        } else {
            lse_push(la, xp, uu);
        }
        acc[L_Q] += xp * xp;
        acc[L_R] += xp * uu;
    };
    for (; i < hi; i += step) {
        const d2 xv = ldg2<BIG>(P.x, i);
        const d2 uv = (MODE & LM_NOU) ? d2{0.0, 0.0} : ldg2<BIG>(P.u, i);
        const d2 g = (MODE & LM_DIR) ? ldg2<BIG>(P.g, i) : d2{0.0, 0.0};
        double x0 = xv.x, x1 = xv.y, u0 = uv.x, u1 = uv.y;
        one(x0, u0, g.x, false, nullptr, nullptr);
        one(x1, u1, g.y, false, nullptr, nullptr);
        if (MODE & LM_ACCEPT) stg2<BIG>(P.x, i, d2{x0, x1});
        if (MODE & LM_DIR) stg2<BIG>(P.u, i, d2{u0, u1});
    }
    if ((P.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long j = P.n - 1;
        double x = P.x[j], u = (MODE & LM_NOU) ? 0.0 : P.u[j];
        const double g = (MODE & LM_DIR) ? P.g[j] : 0.0;
        one(x, u, g, true, P.x + j, P.u + j);
    }
    if (REF) {
        KParams Q; Q.partials = P.partials;
        store_partials(acc, Q);
    } else {
        store_partials_lse(la, acc, P.partials);
    }
}

// The L-BFGS direction pass (k_lbfgs_combine) fused with PHASE 1 OF THE NEXT LINE SEARCH'S FIRST TRIAL: u is in registers
// when it is formed, the first step a₀ = a* (optim.jl:92) is known beforehand, so one more read stream (x, 8 B/elt) buys
// the statistics of xp = x + a₀·u that a k_lse_stats launch of its own would read x AND u for (16 B/elt), plus its
// reduction launch and two kernel boundaries (round 3; config 4: one launch fewer per outer iteration).
template <bool BIG>
__global__ __launch_bounds__(BLOCK) void k_lbfgs_combine_lse(const GramDirParams P, const double *x, double a_trial) {
    LseAcc la{-INFINITY, 0.0, 0.0};
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
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
    auto stats = [&](double xv, double g, double r) {
        acc[S_GU] = dsum(acc[S_GU], g, r);   // as k_lbfgs_combine forms them: the direction sums do not depend on the fusion
        acc[S_UU] = dsum(acc[S_UU], r, r);
        const double xp = xv + a_trial * r;
        lse_push(la, xp, r);
        acc[L_Q] += xp * xp;
        acc[L_R] += xp * r;
    };
    for (; i < hi; i += step) {
        const d2 g = ldg2<BIG>(P.g, i), xv = ldg2<BIG>(x, i);
        const d2 r = lbfgs_combine_pair<BIG>(P, i, g);
        stg2<BIG>(P.u, i, r);
        stats(xv.x, g.x, r.x);
        stats(xv.y, g.y, r.y);
    }
    if ((P.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long e = P.n - 1;
        const double g = P.g[e];
        const double r = lbfgs_combine_one(P, e, g);
        P.u[e] = r;
        stats(x[e], g, r);
    }
    store_partials_lse(la, acc, P.partials);
}

// The Gram push of L-BFGS with the gradient of the accepted trial FORMED IN THE PASS (the log-sum-exp form of
// k_lbfgs_push_gram, cgo_kernels.hip.hpp): g⁺_i = exp(xp_i − M)/S + λ·xp_i, xp = x + a·u — k_lse_grad's expression, bit for
// bit, with the statistics (M, S) of the accepted trial — instead of a k_lse_grad launch that writes it and a push that reads
// it back: that launch, its 24 B/element, its reduction launch and two kernel boundaries go.  x advances OUT OF PLACE (L.xo):
// optim.jl:108-121 must be able to return the last good iterate when ‖g⁺‖ is not finite, so the host swaps x / g only after
// that test (lbfgs_push_commit).  Σ g⁺² travels in row slot GRAM_GTGT.
// Every wave of the wave-split push looks at every element; so that xp, exp and the FP64 division (≈ 35 instructions) are
// formed ONCE per element, a workgroup walks FOUR trips (4 × 64 element pairs) per round: wave w forms xp, g⁺, s, y of trip w,
// does that trip's stores and its "new pair" sums, and leaves g⁺ in LDS; after ONE barrier (two buffers on the round's
// parity) every wave runs its own stored pairs over the four trips with g⁺ from LDS (first form, every wave for itself:
// 424–438 µs at n = 1e7; this one 469 vs 478 µs on a slower box, the plain push 409 / 451 µs).  Same sums, same owners per row
// slot; the new-pair sums of the four waves are added in wave order at the end.
// Since the one-ring-pass iteration (k_lbfgs_combine_spec below) this is the push of the iterations whose line search did
// NOT accept its first trial.
template <bool BIG>
__global__ __launch_bounds__(BLOCK) void k_lbfgs_push_gram_lse(const GramPushParams P, const GramLseParams L) {
    constexpr int T = BLOCK / 64;
    __shared__ d2 gts[2][T][64];
    __shared__ double bs[T][5];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double base[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double acc[GRAM_PER_WAVE][5];
#pragma unroll
    for (int l = 0; l < GRAM_PER_WAVE; ++l)
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[l][q] = 0.0;
    const long long n2 = P.n >> 1;
    long long i0, hi, step;
    if (BIG) {
        const long long per = big_chunk_pairs(n2, gridDim.x);
        i0 = per * blockIdx.x;
        hi = (i0 + per < n2) ? i0 + per : n2;
        step = 64 * T;
    } else {
        i0 = (long long)blockIdx.x * (64 * T);
        hi = n2;
        step = (long long)gridDim.x * (64 * T);
    }
    double *sn = P.S + (size_t)P.slot * ring_ld(P.n), *yn = P.Y + (size_t)P.slot * ring_ld(P.n);
    const double *Sj[GRAM_PER_WAVE], *Yj[GRAM_PER_WAVE];
    bool on[GRAM_PER_WAVE];
#pragma unroll
    for (int l = 0; l < GRAM_PER_WAVE; ++l) {
        const int j = l * 4 + wave;
        on[l] = j < P.count;
        const int slot = on[l] ? P.prev[j] : 0;
        Sj[l] = P.S + (size_t)slot * ring_ld(P.n);
        Yj[l] = P.Y + (size_t)slot * ring_ld(P.n);
    }
    auto lse_gt = [&](double xp) { return exp(xp - L.M) / L.S + L.lambda * xp; };   // k_lse_grad's expression
    int buf = 0;
    for (; i0 < hi; i0 += step, buf ^= 1) {
        {   // wave w: trip w of the round — the state update, g⁺, the candidate pair and its four sums
            const long long i = i0 + 64 * wave + lane;
            if (i < hi) {
                const d2 x = ldg2<BIG>(P.x, i), u = ldg2<false>(P.u, i), g = ldg2<false>(P.g, i);
                d2 xn, gt, s, y;
                xn.x = x.x + P.a * u.x; xn.y = x.y + P.a * u.y;
                gt.x = lse_gt(xn.x); gt.y = lse_gt(xn.y);
                s.x = P.a_s * u.x; s.y = P.a_s * u.y;
                y.x = gt.x - g.x; y.y = gt.y - g.y;
                gts[buf][wave][lane] = gt;
                stg2<BIG>(L.xo, i, xn);
                stg2<BIG>(L.gt_out, i, gt);
                stg2<BIG>(sn, i, s);
                stg2<BIG>(yn, i, y);
                base[0] = dsum(base[0], s.x, y.x);  base[0] = dsum(base[0], s.y, y.y);
                base[1] = dsum(base[1], y.x, y.x);  base[1] = dsum(base[1], y.y, y.y);
                base[2] = dsum(base[2], s.x, gt.x); base[2] = dsum(base[2], s.y, gt.y);
                base[3] = dsum(base[3], y.x, gt.x); base[3] = dsum(base[3], y.y, gt.y);
                base[4] = dsum(base[4], gt.x, gt.x); base[4] = dsum(base[4], gt.y, gt.y);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < T; ++k) {   // every wave: its stored pairs against the four trips
            const long long i = i0 + 64 * k + lane;
            if (i < hi) {
                const d2 u = ldg2<false>(P.u, i), g = ldg2<false>(P.g, i);
                d2 sj[GRAM_PER_WAVE], yj[GRAM_PER_WAVE];
#pragma unroll
                for (int l = 0; l < GRAM_PER_WAVE; ++l)
                    if (on[l]) { sj[l] = ldg2<BIG>(Sj[l], i); yj[l] = ldg2<BIG>(Yj[l], i); }
                const d2 gt = gts[buf][k][lane];
                d2 s, y;
                s.x = P.a_s * u.x; s.y = P.a_s * u.y;
                y.x = gt.x - g.x; y.y = gt.y - g.y;
#pragma unroll
                for (int l = 0; l < GRAM_PER_WAVE; ++l) {
                    if (on[l]) {
                        acc[l][0] = dsum(acc[l][0], sj[l].x, gt.x); acc[l][0] = dsum(acc[l][0], sj[l].y, gt.y);
                        acc[l][1] = dsum(acc[l][1], yj[l].x, gt.x); acc[l][1] = dsum(acc[l][1], yj[l].y, gt.y);
                        acc[l][2] = dsum(acc[l][2], sj[l].x, y.x);  acc[l][2] = dsum(acc[l][2], sj[l].y, y.y);
                        acc[l][3] = dsum(acc[l][3], yj[l].x, s.x);  acc[l][3] = dsum(acc[l][3], yj[l].y, s.y);
                        acc[l][4] = dsum(acc[l][4], yj[l].x, y.x);  acc[l][4] = dsum(acc[l][4], yj[l].y, y.y);
                    }
                }
            }
        }
    }
    if ((P.n & 1) && blockIdx.x == 0 && lane == 0) {  // odd tail element: lane 0 of every wave, its own pairs
        const long long e = P.n - 1;
        const double u = P.u[e], s = P.a_s * u, xe = P.x[e] + P.a * u, gt = lse_gt(xe), y = gt - P.g[e];
        if (wave == 0) {
            L.xo[e] = xe; L.gt_out[e] = gt; sn[e] = s; yn[e] = y;
            base[0] = dsum(base[0], s, y); base[1] = dsum(base[1], y, y); base[2] = dsum(base[2], s, gt); base[3] = dsum(base[3], y, gt);
            base[4] = dsum(base[4], gt, gt);
        }
#pragma unroll
        for (int l = 0; l < GRAM_PER_WAVE; ++l) {
            if (on[l]) {
                const double sje = Sj[l][e], yje = Yj[l][e];
                acc[l][0] = dsum(acc[l][0], sje, gt); acc[l][1] = dsum(acc[l][1], yje, gt); acc[l][2] = dsum(acc[l][2], sje, y);
                acc[l][3] = dsum(acc[l][3], yje, s);  acc[l][4] = dsum(acc[l][4], yje, y);
            }
        }
    }
    double *row = P.partials + (size_t)blockIdx.x * NG;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const double v = wave_sum(base[q]);
        if (lane == 0) bs[wave][q] = v;
    }
#pragma unroll
    for (int l = 0; l < GRAM_PER_WAVE; ++l) {
        const int j = l * 4 + wave;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const double v = wave_sum(acc[l][q]);
            if (lane == 0 && j < GRAM_MAXC && 4 + 5 * j + q != GRAM_GTGT) row[4 + 5 * j + q] = on[l] ? v : 0.0;
        }
    }
    __syncthreads();
    if (tid < 5) {
        double v = bs[0][tid];
#pragma unroll
        for (int w = 1; w < T; ++w) v += bs[w][tid];
        row[tid < 4 ? tid : GRAM_GTGT] = v;
    }
}

// (The one-ring-pass L-BFGS kernels — k_lbfgs_combine_spec, k_lbfgs_push_lite, for log-sum-exp AND the element-wise objectives —
// live in cgo_kernels.hip.hpp: the run-time compiled user objectives instantiate them too.)

// phase 2: g⁺_i = exp(xp_i − M)/S + λ·xp_i, plus ‖g⁺‖² and the getβ partial sums.
// INIT: xp = x (no u), also writes u = −g⁺.
template <bool BETA, bool INIT, bool BIG>
__global__ __launch_bounds__(BLOCK) void k_lse_grad(const LseParams P) {
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.0;
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
    auto one = [&](double x, double u, double g, double &gt_out, double &u_out) {
        const double xp = INIT ? x : (x + P.a_trial * u);
        const double gt = exp(xp - P.M) / P.S + P.lambda * xp;
        gt_out = gt;
        acc[S_GTGT] += gt * gt;
        if (INIT) { u_out = -gt; return; }
        acc[S_GTU] += gt * u;
        if (BETA) {
            const double y = gt - g;
            acc[S_GTG] += gt * g; acc[S_YY] += y * y; acc[S_UY] += u * y; acc[S_YGT] += y * gt;
        }
    };
    for (; i < hi; i += step) {
        const d2 x = ldg2<BIG>(P.x, i);
        const d2 u = INIT ? d2{0.0, 0.0} : ldg2<BIG>(P.u, i);
        const d2 g = (BETA && !INIT) ? ldg2<BIG>(P.g, i) : d2{0.0, 0.0};
        double gt0, gt1, un0 = 0.0, un1 = 0.0;
        one(x.x, u.x, g.x, gt0, un0);
        one(x.y, u.y, g.y, gt1, un1);
        stg2<BIG>(P.gt, i, d2{gt0, gt1});
        if (INIT) stg2<BIG>(P.u, i, d2{un0, un1});
    }
    if ((P.n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long j = P.n - 1;
        double gt, un = 0.0;
        one(P.x[j], INIT ? 0.0 : P.u[j], (BETA && !INIT) ? P.g[j] : 0.0, gt, un);
        P.gt[j] = gt;
        if (INIT) P.u[j] = un;
    }
    KParams Q; Q.partials = P.partials;
    store_partials(acc, Q);
}

}  // namespace dev
}  // namespace cgo
