// cgo_kernels_chain.hip.hpp — the CHAINED Rosenbrock objective as a device STENCIL objective.
//
//   f(x) = Σ_{i=0}^{N−2} (1 − x_i)² + 100 (x_{i+1} − x_i²)²            examples/helpers/test_funcs.jl:50-57 (value;
//   ∂f/∂x_k = [k ≥ 1] 200 t_{k−1} + [k ≤ N−2] (−2(1 − x_k) − 400 x_k t_k),   t_i = x_{i+1} − x_i²     BASELINE config 1's
//                                                                        "Rosenbrock n = 1000" in its chained form)
//
// Unlike every other device objective ∇f_k couples x_{k−1}, x_k, x_{k+1}, so the gradient-free idea of
// cgo_kernels_cg.hip.hpp — recompute g = ∇f(x) in registers instead of keeping it in HBM — needs neighbours:
//   * a lane owns one pair (x_{2p}, x_{2p+1}) and loads the pairs left and right of it as well (three 16-B loads per
//     vector, the two extra ones served by L1/L2: HBM traffic stays 1×);
//   * the fused accept + direction + trial launch needs x_new on a window of six elements, g(x_new), u_new and the trial
//     point on the middle four, g⁺ on the own two — everything is recomputed redundantly by the neighbouring lanes
//     with the SAME expressions, hence bit-identical (≈ 3× the flops of the paired form, still far from binding);
//   * x and u are updated OUT OF PLACE (xo/uo, buffers swapped by the backend after the launch): an in-place update
//     would let a neighbouring workgroup read an element that has already been advanced;
//   * across shard boundaries (SURVEY.md §8e "Partitioning") the window reaches two elements into the neighbour rank: the
//     launch takes those four values of x and u per side as ARGUMENTS (halo) and leaves the new values of its own two
//     edge elements per side in slots 10–17 of its reduction row, so that they travel in the per-launch scalar block every
//     rank receives anyway — no extra message, no P2P copy.  has_left/has_right = 0 at the ends of the global vector.
//
// One or three trial points per launch (row width 24: ten sums + eight edge values, padded; or 32: the 23 sums of a
// 3-point row, then the edge values): the speculative points of cgo_kernels_cg.hip.hpp cost a stencil launch only more
// of the same window arithmetic.  Arithmetic is unfused and in the oracle's order (oracle/cgo_oracle.c orc_fdf_rosenbrock_chained: g_k = (0 + 200 t_{k−1}) + (−2 t2_k − 400 (x_k t_k))).
#pragma once

#include "cgo_kernels_cg.hip.hpp"

namespace cgo {
namespace dev {

constexpr int NRC = 24;        // row width of a 1-point chain launch (shares k_finalize_t<24, …> with the 3-point k_cg rows)
constexpr int RC_EDGE = 10;    // slots 10–13: x_new, u_new of the first two elements; 14–17: of the last two
constexpr int NRC3 = 32;       // row width of a 3-point chain launch: sums as in a 3-point k_cg row (0–22) …
constexpr int RC3_EDGE = 24;   // … edge values in slots 24–31
constexpr int R_EDGES = 512;   // mode bit: publish the edge values of the CURRENT x, u only (after set_x0)
template <int NPTS> struct ChainRow {
    static constexpr int W = NPTS == 1 ? NRC : NRC3, EDGE = NPTS == 1 ? RC_EDGE : RC3_EDGE;
    static constexpr int GU = RS_PER_POINT * NPTS, UU = GU + 1;   // Σ g·u_new, Σ u_new·u_new behind the trial sums
};

struct ChainParams {
    const double *x; const double *u;
    double *xo; double *uo;       // where updated x / u go (never the buffers being read)
    double *gout;
    long long n;                  // local length, padded to even: with `odd` the last element is a phantom that does not exist
    int odd;                      // the global vector ends in a single element (test_funcs.jl:50-57 loops 1:N−1 for ANY N): odd N
    double a_acc, beta;
    double a[3];                  // trial steps (NPTS of them; R_GRADT: a[0])
    double *partials;
    double hxl[2], hul[2], hxr[2], hur[2];   // x, u of the two elements left of local 0 / right of local n−1
    int has_left, has_right;      // 0: that side is the end of the global vector
    Tail tail;
};

// ∇f_k from x_{k−1}, x_k, x_{k+1}; em / ep: does element k−1 / k+1 exist (globally)
__device__ inline double chain_grad(double xm, double x0, double xp, bool em, bool ep) {
    double g = 0.0;
    if (em) { const double t1 = x0 - xm * xm; g = g + 200.0 * t1; }
    if (ep) { const double t2 = 1.0 - x0, t1 = xp - x0 * x0; g = g + (-2.0 * t2 - 400.0 * (x0 * t1)); }
    return g;
}
// term k of f (exists iff element k+1 does)
__device__ inline double chain_term(double x0, double xp) {
    const double t2 = 1.0 - x0, t1 = xp - x0 * x0;
    return t2 * t2 + 100.0 * (t1 * t1);
}

// The stencil arithmetic of ONE pair on its window of six elements — shared by the launch-per-trial kernel below and by the
// resident form (cgo_kernels_resident.hip.hpp): X/U[0..1] = the pair to the left (or the halo), [2..3] = the own pair,
// [4..5] = the pair to the right; E[k]: does element k exist.  Leaves the own pair's new x / u in xn2 / un2 (what a launch
// that writes them stores), the gradient asked for (R_GRAD: ∇f(x), R_GRADT: ∇f(x + a₀u)) in gout2, and adds the pair's terms
// to acc (row layout of a k_cg row: 7 sums per trial point, then Σ g·u_new, Σ u_new·u_new).
template <int MODE, int NPTS, int W>
__device__ inline void chain_window(const double (&X)[6], const double (&U)[6], const bool (&E)[6], double a_acc, double beta,
                                    const double (&a)[3], double (&acc)[W], d2 &xn2, d2 &un2, d2 &gout2) {
    constexpr int GU = RS_PER_POINT * NPTS, UU = GU + 1;
    double xn[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) xn[k] = (MODE & R_ACCEPT) ? X[k] + a_acc * U[k] : X[k];   // optim.jl:136,140
    double g[6], un[6];   // indices 1..4 are used
#pragma unroll
    for (int k = 1; k <= 4; ++k) g[k] = E[k] ? chain_grad(xn[k - 1], xn[k], xn[k + 1], E[k - 1], E[k + 1]) : 0.0;
#pragma unroll
    for (int k = 1; k <= 4; ++k) {
        if (MODE & R_DIR) un[k] = -g[k] + beta * U[k];                // cg_flavours.jl:10-12
        else if (MODE & (R_INIT | R_RESET)) un[k] = -g[k];              // cg_flavours.jl:29, wolfe.jl:129
        else un[k] = U[k];
        if (!E[k]) un[k] = 0.0;
    }
    if (MODE & R_INIT) {
#pragma unroll
        for (int k = 2; k <= 3; ++k) {
            if (!E[k]) continue;
            if (E[k + 1]) acc[RS_F] += chain_term(xn[k], xn[k + 1]);
            acc[RS_GTGT] = dsum(acc[RS_GTGT], g[k], g[k]);
        }
    }
    if (MODE & (R_DIR | R_RESET)) {
#pragma unroll
        for (int k = 2; k <= 3; ++k) {
            if (!E[k]) continue;
            acc[GU] = dsum(acc[GU], g[k], un[k]);      // Σ g·u_new   (behind the trial sums, as in a k_cg row)
            acc[UU] = dsum(acc[UU], un[k], un[k]);     // Σ u_new·u_new
        }
    }
    if (MODE & R_UPG) {
#pragma unroll
        for (int k = 2; k <= 3; ++k) { if (!E[k]) continue; const double t = U[k] + g[k]; acc[UU] = dsum(acc[UU], t, t); }
    }
    if (MODE & R_GRAD) gout2 = d2{g[2], g[3]};
    if (MODE & (R_TRIAL | R_GRADT)) {
#pragma unroll
        for (int j = 0; j < NPTS; ++j) {
            double xp[6], gt[4];
#pragma unroll
            for (int k = 1; k <= 4; ++k) xp[k] = xn[k] + a[j] * un[k];     // cg_utils.jl:14-16
#pragma unroll
            for (int k = 2; k <= 3; ++k) gt[k] = E[k] ? chain_grad(xp[k - 1], xp[k], xp[k + 1], E[k - 1], E[k + 1]) : 0.0;
            if (MODE & R_GRADT) gout2 = d2{gt[2], gt[3]};
            if (MODE & R_TRIAL) {
                const int b = RS_PER_POINT * j;   // point j's seven sums (row layout of cgo_kernels_cg.hip.hpp)
#pragma unroll
                for (int k = 2; k <= 3; ++k) {
                    if (!E[k]) continue;
                    if (E[k + 1]) acc[b + RS_F] += chain_term(xp[k], xp[k + 1]);
                    const double y = gt[k] - g[k];
                    acc[b + RS_GTU] = dsum(acc[b + RS_GTU], gt[k], un[k]);
                    acc[b + RS_GTGT] = dsum(acc[b + RS_GTGT], gt[k], gt[k]);
                    acc[b + RS_GTG] = dsum(acc[b + RS_GTG], gt[k], g[k]);
                    acc[b + RS_YY] = dsum(acc[b + RS_YY], y, y);
                    acc[b + RS_UY] = dsum(acc[b + RS_UY], un[k], y);
                    acc[b + RS_YGT] = dsum(acc[b + RS_YGT], y, gt[k]);
                }
            }
        }
    }
    xn2 = d2{xn[2], xn[3]};
    un2 = d2{un[2], un[3]};
}

template <int MODE, int NPTS, bool BIG>
__global__ __launch_bounds__(BLOCK) void k_chain(const ChainParams P) {
    constexpr int W = ChainRow<NPTS>::W, EDGE = ChainRow<NPTS>::EDGE;
    double acc[W];
#pragma unroll
    for (int s = 0; s < W; ++s) acc[s] = 0.0;
    constexpr bool need_u = (MODE & (R_ACCEPT | R_DIR | R_TRIAL | R_UPG | R_GRADT | R_EDGES)) != 0;
    constexpr bool wr_x = (MODE & R_ACCEPT) != 0;
    constexpr bool wr_u = (MODE & (R_DIR | R_INIT | R_RESET)) != 0;
    const long long n2 = P.n >> 1;
    long long i, hi, step;
    if (BIG) {
        const long long per = (n2 + gridDim.x - 1) / gridDim.x;
        i = per * blockIdx.x + threadIdx.x;
        hi = (per * blockIdx.x + per < n2) ? per * blockIdx.x + per : n2;
        step = BLOCK;
    } else {
        i = (long long)blockIdx.x * BLOCK + threadIdx.x;
        hi = n2;
        step = (long long)gridDim.x * BLOCK;
    }
    const double a3[3] = {P.a[0], P.a[1], P.a[2]};
    for (; i < hi; i += step) {
        // window of six elements: X[0..1] = pair i−1 (or the left halo), X[2..3] = own pair, X[4..5] = pair i+1 (or right halo)
        double X[6], U[6];
        bool E[6];
        const bool first = (i == 0), last = (i == n2 - 1);
        {
            const d2 c = ldg2<false>(P.x, i);
            X[2] = c.x; X[3] = c.y;
            if (!first) { const d2 l = ldg2<false>(P.x, i - 1); X[0] = l.x; X[1] = l.y; } else { X[0] = P.hxl[0]; X[1] = P.hxl[1]; }
            if (!last) { const d2 r = ldg2<false>(P.x, i + 1); X[4] = r.x; X[5] = r.y; } else { X[4] = P.hxr[0]; X[5] = P.hxr[1]; }
        }
        if (need_u) {
            const d2 c = ldg2<false>(P.u, i);
            U[2] = c.x; U[3] = c.y;
            if (!first) { const d2 l = ldg2<false>(P.u, i - 1); U[0] = l.x; U[1] = l.y; } else { U[0] = P.hul[0]; U[1] = P.hul[1]; }
            if (!last) { const d2 r = ldg2<false>(P.u, i + 1); U[4] = r.x; U[5] = r.y; } else { U[4] = P.hur[0]; U[5] = P.hur[1]; }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) U[k] = 0.0;
        }
        E[0] = E[1] = !first || P.has_left != 0;
        E[2] = true;
        E[3] = !(last && P.odd != 0);   // odd N: the last pair's second element is padding (x = u = 0, contributes nothing)
        E[4] = E[5] = !last || P.has_right != 0;
        if (P.odd != 0 && i + 1 == n2 - 1) E[5] = false;   // … and it is the right neighbour pair's second element for the pair before

        if (MODE & R_EDGES) {   // nothing to compute: the edge values of the state as it is
            if (first) { acc[EDGE + 0] = X[2]; acc[EDGE + 1] = X[3]; acc[EDGE + 2] = U[2]; acc[EDGE + 3] = U[3]; }
            if (last) { acc[EDGE + 4] = X[2]; acc[EDGE + 5] = X[3]; acc[EDGE + 6] = U[2]; acc[EDGE + 7] = U[3]; }
            continue;
        }
        d2 xn2, un2, g2 = d2{0.0, 0.0};
        chain_window<MODE, NPTS, W>(X, U, E, P.a_acc, P.beta, a3, acc, xn2, un2, g2);
        if (MODE & (R_GRAD | R_GRADT)) stg2<false>(P.gout, i, g2);
        if (wr_x) stg2<false>(P.xo, i, xn2);
        if (wr_u) stg2<false>(P.uo, i, un2);
        // this rank's edge values AFTER the launch, for the neighbours' next window (exactly one lane owns each)
        if (first) { acc[EDGE + 0] = xn2.x; acc[EDGE + 1] = xn2.y; acc[EDGE + 2] = un2.x; acc[EDGE + 3] = un2.y; }
        if (last) { acc[EDGE + 4] = xn2.x; acc[EDGE + 5] = xn2.y; acc[EDGE + 6] = un2.x; acc[EDGE + 7] = un2.y; }
    }
    if (MODE == R_ACCEPT || MODE == R_GRAD || MODE == R_GRADT) return;   // no sums (an accept-only launch ends the solve)
    store_partials_n<W>(acc, P.partials, P.tail);
}

}  // namespace dev
}  // namespace cgo
