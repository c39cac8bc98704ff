// cgo_ctl.hpp — scalar decisions shared by the host engine and the on-device controller.
//
// One source of truth, compiled for the host (cgo_engine.cpp, tests/hostsim) and for gfx950
// (k_ctl in cgo_hip_backend.hip): the β formulas on reduced sums, the first step / candidate
// steps of a line search, the Wolfe tests, and `ctl_step` — the decision "is the first trial of
// this line search accepted, and if so what does the next fused launch need?".  With it the GPU
// can run a streak of outer iterations whose line search succeeds at its first trial WITHOUT
// any host round trip: launch k+1 reads its scalars (a*, β, the three
// trial steps) from device memory written by the controller after launch k.
//
// IEEE-754 basic operations and sqrt are correctly rounded on both sides and nothing here is
// contracted (-ffp-contract=off), so host and device take identical decisions.
#pragma once

#include <stdint.h>

#include "../../include/cgo.h"

#if defined(__HIPCC__) || defined(__HIP__) || defined(CGO_RTC)
#define CGO_HD __host__ __device__
#else
#define CGO_HD
#endif

namespace cgo {

// the seven trial sums of one point + direction sums, as the kernels deliver them
struct TrialSums { double f, gtu, gtgt, gtg, yy, uy, ygt; };

CGO_HD inline bool hd_isfinite(double v) { return __builtin_isfinite(v); }
CGO_HD inline bool hd_isnan(double v) { return __builtin_isnan(v); }
// Base.max / Base.min propagate NaN (cg_flavours.jl:68, wolfe.jl:243,247)
CGO_HD inline double jl_max(double a, double b) { return (hd_isnan(a) || hd_isnan(b)) ? __builtin_nan("") : (a > b ? a : b); }
CGO_HD inline double jl_min(double a, double b) { return (hd_isnan(a) || hd_isnan(b)) ? __builtin_nan("") : (a < b ? a : b); }

// LinearAlgebra.norm from Σv²: sqrt(Σv²) is the true 2-norm unless the squares over- or underflowed; outside
// [1e-280, 1e300] the caller must supply the scaled form max|v|·sqrt(Σ(v/max)²) instead (Solver::robust_norm).
CGO_HD inline bool sumsq_in_range(double ss) { return ss >= 1e-280 && ss <= 1e300; }

struct BetaNorms {   // norm(u), norm(y), norm(g_next) as LinearAlgebra.norm returns them (cg_flavours.jl:65,140)
    double u, y, gt;
};
CGO_HD inline BetaNorms beta_norms_fast(const TrialSums &t, double uu_old) {
    BetaNorms n; n.u = __builtin_sqrt(uu_old); n.y = __builtin_sqrt(t.yy); n.gt = __builtin_sqrt(t.gtgt); return n;
}
// which of them a flavour reads, and whether the fast form is the true norm for all of those
CGO_HD inline bool beta_norms_fast_ok(int kind, const TrialSums &t, double uu_old) {
    if (kind == CGO_BETA_YUAN_WANG_SHENG) return sumsq_in_range(uu_old) && sumsq_in_range(t.yy);
    if (kind == CGO_BETA_SALLEH_ALHAWARAT) return sumsq_in_range(t.gtgt);
    return true;
}

// getβ(β_config, g_next, g, u) evaluated on the one-pass partial sums (cg_flavours.jl:46-170).
// gu_old = u·g (the dϕ₀ of the line search just finished), gg_old = g·g, uu_old = u·u; nrm: see BetaNorms.
CGO_HD inline double beta_from_sums(int kind, double mu, const TrialSums &t, double gu_old, double gg_old,
                                    double uu_old, const BetaNorms &nrm) {
    switch (kind) {
    case CGO_BETA_HAGER_ZHANG: {  // cg_flavours.jl:96-105, Σ(y−m·u)(g⁺/R) expanded on the sums
        const double R = t.uy;
        const double m = 2 * t.yy / R;
        return (t.ygt - m * t.gtu) / R;
    }
    case CGO_BETA_YUAN_WANG_SHENG: {  // cg_flavours.jl:63-76
        const double R1 = mu * nrm.u * nrm.y;   // μ·norm(u)·norm(y): the TRUE norms (scaled form in extreme ranges)
        const double R2 = t.uy;
        const double R3 = 2 * t.yy * t.gtu / t.ygt;
        const double R = jl_max(jl_max(R1, R2), R3);
        const double m = 2 * t.yy / R;
        return (t.ygt - m * t.gtu) / R;
    }
    case CGO_BETA_SALLEH_ALHAWARAT: {  // cg_flavours.jl:140-150
        const double norm_sq = nrm.gt * nrm.gt;     // norm(g_next)^2: the true norm, then squared
        if (norm_sq > t.gtg) return (norm_sq - t.gtg) / (t.gtu - gu_old);
        return 0.0;
    }
    case CGO_BETA_LIU_STORREY:  // cg_flavours.jl:164-169
        return t.ygt / (-t.uy);
    case CGO_BETA_HESTENES_STIEFEL:  // cg_flavours.jl:121-126 (commented there)
        return t.ygt / t.uy;
    case CGO_BETA_POLAK_RIBIERE:
        return t.ygt / gg_old;
    case CGO_BETA_DAI_YUAN:
        return t.gtgt / t.uy;
    case CGO_BETA_BROYDEN_FAMILY:  // qn_flavours.jl:70-90: B_new = B = I up to rounding ⇒ u = B\(−g) = −g
        return 0.0;
    default:
        return __builtin_nan("");
    }
}

// first step of a line search from the previous accepted step (optim.jl:92)
CGO_HD inline double ls_first_step(const cgo_ls_config &ls, double a_initial) {
    if (ls.kind == CGO_LS_BACKTRACKING) return a_initial;  // geometric.jl:48-56 (non-finite → |ϕ₀|/u·u, later)
    if (ls.kind == CGO_LS_STRONG_WOLFE_BISECTION) {
        if (!(0.0 < a_initial && hd_isfinite(a_initial))) return 1.0;  // nocedal.jl:49-52
        return a_initial;
    }
    if (!(ls.max_step_size > a_initial && a_initial > 0.0))            // wolfe.jl:30-32
        return jl_min(1.0, ls.max_step_size / 2);
    return a_initial;
}

// The two steps a line search can request right after its FIRST trial at a0 (lb/lo = 0):
//   StrongWolfeBisection: zoom midpoint (0+a0)/2 | extrapolation (a0·growth + a0)/2   nocedal.jl:81-150,186
//   WolfeBisection:       (0+a0)/2 | 2·a0                                               wolfe.jl:86-114
//   Backtracking:         a0/ρ | a0·ρ                                                   geometric.jl:7-13,126
CGO_HD inline void ls_first_hints(const cgo_ls_config &ls, double a0, double &h0, double &h1) {
    if (ls.kind == CGO_LS_STRONG_WOLFE_BISECTION) {
        h0 = (0.0 + a0) / 2;
        h1 = (a0 * ls.a_max_growth_factor + a0) / 2;
    } else if (ls.kind == CGO_LS_WOLFE_BISECTION) {
        h0 = (0.0 + a0) / 2;
        h1 = 2.0 * a0;
    } else {
        h0 = a0 / ls.discount_factor;
        h1 = a0 * ls.discount_factor;
    }
}

// requested step + its distinct, finite, positive candidates → pts[0..k)
CGO_HD inline int ls_trial_points(const cgo_ls_config &ls, double a0, bool multi, double (&pts)[3]) {
    // (no run-time index into a local array anywhere on the controller's path: on the device that would put the array —
    // and with it a scratch allocation for EVERY wave of an armed launch — into private memory)
    pts[0] = a0; pts[1] = 0; pts[2] = 0;
    int k = 1;
    if (multi) {
        double h0, h1;
        ls_first_hints(ls, a0, h0, h1);
        if (hd_isfinite(h0) && h0 > 0.0 && h0 != pts[0]) { pts[1] = h0; k = 2; }
        if (hd_isfinite(h1) && h1 > 0.0 && h1 != pts[0] && (k < 2 || h1 != pts[1])) {
            if (k == 1) pts[1] = h1; else pts[2] = h1;
            ++k;
        }
    }
    return k;
}

// 5-point launches add, under each of the two candidates, the grandchild on the side of a0 (the
// previous accepted step is the best guess of the next one, so the bracket usually closes towards
// it): zoom (0,a0)→a0/2→(a0/2+a0)/2 and extrapolation e→zoom(a0,e) midpoint (nocedal.jl:81-150,186);
// bisection halves likewise, or (a0+2a0)/2 after a doubling (wolfe.jl:86-114); a further factor ρ for
// Backtracking (geometric.jl:126).  Pure speculation: a wrong guess costs nothing but a later launch.
// Seven points go one more level down the same two paths.
CGO_HD inline int ls_trial_points_n(const cgo_ls_config &ls, double a0, int maxp, double (&pts)[7]) {
    double h0, h1;
    ls_first_hints(ls, a0, h0, h1);
    double g0, g1, q0, q1;
    if (ls.kind == CGO_LS_BACKTRACKING) {
        g0 = h0 / ls.discount_factor; g1 = h1 * ls.discount_factor;
        q0 = g0 / ls.discount_factor; q1 = g1 * ls.discount_factor;
    } else {
        g0 = (h0 + a0) / 2; g1 = (a0 + h1) / 2;
        q0 = (g0 + a0) / 2; q1 = (a0 + g1) / 2;
    }
    const double cand[6] = {h0, h1, g0, g1, q0, q1};
    pts[0] = a0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 1; j < 7; ++j) pts[j] = 0.0;
    int k = 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < 6; ++q) {   // (every index below is a compile-time constant once the loops are unrolled)
        const double v = cand[q];
        bool ok = k < maxp && hd_isfinite(v) && v > 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < 7; ++j) if (j < k && v == pts[j]) ok = false;
        if (ok) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int j = 1; j < 7; ++j) if (j == k) pts[j] = v;
            ++k;
        }
    }
    return k;
}

// wolfe.jl:219-294
CGO_HD inline void wolfe_tests(const cgo_ls_config &ls, double phi0, double d0, double uu, double phi_a,
                               double dphi_a, double a, bool &ok_large, bool &ok_small) {
    const double c1 = ls.c1, c2 = ls.c2;
    if (ls.cond_kind == CGO_COND_YUAN_WEI_LU) {
        const double d1 = ls.delta1, nu = uu;
        const double rhs1 = phi0 + c1 * a * d0 + a * jl_min(-d1 * d0, c1 * a * nu / 2);
        const double rhs2 = c2 * d0 + jl_min(-d1 * d0, c1 * a * nu);
        ok_large = phi_a <= rhs1;
        ok_small = dphi_a >= rhs2;
    } else {
        ok_large = phi_a <= phi0 + c1 * a * d0;
        ok_small = dphi_a >= c2 * d0;
    }
}

// ---- the two bisection line searches as state machines over an evaluator ----------------------------
// One definition for the host engine (the evaluator launches kernels, cgo_engine.cpp Solver::eval) and for the
// on-device controller (the evaluator only looks trial results up among the points the finished launch evaluated,
// and aborts the search on a miss).  `ev(a, phi, dphi, h1, h2, h3, h4)` returns 0 or an abort code that is passed up;
// h1..h4 are the steps the search can ask for next (speculation hints).
struct LSOut { double phi, a; int64_t evals; int status; };

CGO_HD inline LSOut ls_out(double phi, double a, int64_t evals, int status) {
    LSOut o; o.phi = phi; o.a = a; o.evals = evals; o.status = status; return o;
}

// nocedal.jl:162-209  zoom!
template <class Ev>
CGO_HD inline int ls_zoom_t(const cgo_ls_config &ls, double phi0, double d0, double lo, double hi, double phi_lo,
                            int64_t evals, Ev &ev, LSOut &o) {
    const double c1 = ls.c1, c2 = ls.c2;
    double a = 0, phi = 0, dphi = 0;
    int run_hi = 0, run_lo = 0;   // how many times in a row `hi` / `lo` was the bound that moved (speculation hints only)
    for (int64_t k = 0; k < ls.zoom_max_iters; ++k) {
        a = (lo + hi) / 2;
        // next midpoint: lower | upper half; then the quarter next to `a` on either side.  A zoom that has moved the same
        // bound twice in a row is on a monotone run (strict c2 on a non-quadratic objective: 1, 1/2, 1/4, 1/8, … —
        // profiles/r02_line_search_paths_config1.log): then the next TWO midpoints of that run come first, so that a
        // three-point trial launch walks three levels of it instead of two.  Hints never change a step.
        const double hl = (lo + a) / 2, hu = (a + hi) / 2;
        int rc;
        if (run_hi >= 2) rc = ev(a, phi, dphi, hl, (lo + hl) / 2, hu, (hl + a) / 2);
        else if (run_lo >= 2) rc = ev(a, phi, dphi, hu, (hu + hi) / 2, hl, (a + hu) / 2);
        else rc = ev(a, phi, dphi, hl, hu, (hl + a) / 2, (a + hu) / 2);
        if (rc) return rc;
        ++evals;
        if ((phi > phi0 + c1 * a * d0) || (phi >= phi_lo)) {
            hi = a;
            ++run_hi; run_lo = 0;
            continue;
        }
        if (__builtin_fabs(dphi) <= -c2 * d0) { o = ls_out(phi, a, evals, CGO_SUCCESS); return 0; }
        if (dphi * (hi - lo) >= 0) { hi = lo; run_lo = 0; } else ++run_lo;
        run_hi = 0;
        lo = a;
        phi_lo = phi;
    }
    o = ls_out(phi, a, evals, CGO_ZOOM_MAX_ITERS_REACHED);
    return 0;
}

// nocedal.jl:33-158  linesearch!(info, ::StrongWolfeBisection, …)
template <class Ev>
CGO_HD inline int ls_strong_wolfe_t(const cgo_ls_config &ls, double phi0, double d0, double a_initial, Ev &ev, LSOut &o) {
    const double c1 = ls.c1, c2 = ls.c2;
    double a = ls_first_step(ls, a_initial);
    if (d0 > 0.0) { o = ls_out(phi0, 0.0, 0, CGO_NON_DESCENT_SEARCH_DIRECTION); return 0; }
    double a_prev = 0.0, phi_prev = phi0, phi = phi0, dphi = d0;
    int64_t evals = 0;
    for (int64_t k = 0; k < ls.max_iters; ++k) {
        // next step: first zoom midpoint of (a_prev, a) | extrapolation (a·growth + a)/2; from the third extrapolation in a
        // row on, the next two extrapolations first (a monotone run: 1, 3/2, 9/4, 27/8, … — see ls_zoom_t)
        const double hz = (a_prev + a) / 2, he = (a * ls.a_max_growth_factor + a) / 2;
        if (int rc = (k >= 3) ? ev(a, phi, dphi, he, (he * ls.a_max_growth_factor + he) / 2, hz, (a + he) / 2)
                              : ev(a, phi, dphi, hz, he, (hz + a) / 2, (a + he) / 2)) return rc;
        ++evals;
        const bool too_high = phi > phi0 + c1 * a * d0;
        const bool not_lower = phi >= phi_prev;
        if (too_high || (not_lower && k > 0)) return ls_zoom_t(ls, phi0, d0, a_prev, a, phi_prev, evals, ev, o);
        if (__builtin_fabs(dphi) <= -c2 * d0) { o = ls_out(phi, a, evals, CGO_SUCCESS); return 0; }
        if (dphi >= 0) return ls_zoom_t(ls, phi0, d0, a, a_prev, phi, evals, ev, o);
        a_prev = a;
        phi_prev = phi;
        const double a_max = a * ls.a_max_growth_factor;
        if (a > a_max) { o = ls_out(phi, a, evals, CGO_LINESEARCH_A_MAX_OVERFLOW); return 0; }
        a = (a_max + a) / 2;
    }
    o = ls_out(phi, a, evals, CGO_LINESEARCH_MAX_ITERS_REACHED);
    return 0;
}

// wolfe.jl:171-207  findfeasiblestepsize!  (reduction_factor fixed at 0.5 by the caller, wolfe.jl:23)
template <class Ev>
CGO_HD inline int ls_find_feasible_t(const cgo_ls_config &ls, double &a, double lb, int64_t &evals, double &phi, double &dphi,
                                     int &flag, Ev &ev, double h1, double h2, double h3, double h4) {
    const double nan = __builtin_nan("");
    if (lb > a) {
        phi = 0.0; dphi = 0.0;
        flag = CGO_BISECTION_LOWER_BOUND_LARGER_THAN_PROPOSED_STEP;
        return 0;
    }
    if (int rc = ev(a, phi, dphi, h1, h2, h3, h4)) return rc;
    ++evals;
    for (int64_t iter = 1; a > lb && iter < ls.feasibility_max_iters; ++iter) {
        if (hd_isfinite(phi) && hd_isfinite(dphi)) { flag = CGO_FEASIBLE; return 0; }
        a = a * 0.5;
        if (int rc = ev(a, phi, dphi, nan, nan, nan, nan)) return rc;
        ++evals;
    }
    flag = CGO_INFEASIBLE;
    return 0;
}

// wolfe.jl:13-165  linesearch!(info, ::WolfeBisection, …).  `uu` = dot(u,u) (YuanWeiLuWolfe re-reads it on every
// check, wolfe.jl:240) is updated when the search resets the direction.  `bk` supplies the two vector operations of
// the bracket-collapse branch (wolfe.jl:122-133): bk.is_neg_grad(bool&) — is norm(u + df_x) == 0 — and
// bk.reset_dir(double &uu) — u = −df_x; both return 0 or an abort code.
template <class Ev, class Bk>
CGO_HD inline int ls_wolfe_bisection_t(const cgo_ls_config &ls, double phi0, double d0, double &uu, double a_initial,
                                       Ev &ev, Bk &bk, LSOut &o) {
    a_initial = ls_first_step(ls, a_initial);
    if (!hd_isfinite(phi0)) { o = ls_out(phi0, 0.0, 0, CGO_ACCEPTED_NON_FINITE_ITERATE); return 0; }
    if (d0 > 0.0) { o = ls_out(phi0, 0.0, 0, CGO_NON_DESCENT_SEARCH_DIRECTION); return 0; }
    const double inf = __builtin_inf();
    double a = a_initial, lb = 0.0, ub = inf, phi = 0, dphi = 0;
    int64_t evals = 0;
    int flag = 0;
    if (int rc = ls_find_feasible_t(ls, a, 0.0, evals, phi, dphi, flag, ev, (lb + a) / 2, 2.0 * a, ((lb + a) / 2 + a) / 2,
                                    (a + 2.0 * a) / 2)) return rc;
    if (flag != CGO_FEASIBLE) { o = ls_out(phi0, 0.0, 0, CGO_CANNOT_FIND_INITIAL_FEASIBLE_STEP); return 0; }
    for (int64_t k = 0; k < ls.max_iters; ++k) {
        bool ok_large, ok_small;
        wolfe_tests(ls, phi0, d0, uu, phi, dphi, a, ok_large, ok_small);
        if (ok_large && ok_small) { o = ls_out(phi, a, evals, CGO_SUCCESS); return 0; }
        if (!ok_large) {            // step too long: shrink the bracket from above
            ub = a;
            a = (lb + ub) / 2;
        } else {                    // step too short
            lb = a;
            if (!hd_isfinite(ub)) {
                a = 2.0 * a;        // growth_factor, wolfe.jl:24,102
                if (a > ls.max_step_size) { o = ls_out(phi0, 0.0, 0, CGO_MAX_STEP_LENGTH_REACHED); return 0; }
            } else {
                a = (lb + ub) / 2;
            }
        }
        if (!(lb < a && a < ub)) {  // bracket collapsed, wolfe.jl:122-133
            // `!isapprox(norm(u+df_x), 0)`: with default tolerances this is norm ≠ 0 exactly.
            bool is_neg_grad = false;
            if (int rc = bk.is_neg_grad(is_neg_grad)) return rc;
            if (!is_neg_grad) {     // restart from steepest descent; dϕ₀ is NOT recomputed (wolfe.jl:125-129)
                lb = 0.0; ub = inf; a = a_initial;
                if (int rc = bk.reset_dir(uu)) return rc;
            }
            // else: wolfe.jl:131 builds a tuple and drops it (missing `return`) → falls through
        }
        // whatever this trial yields, the next step is the lower half | the upper half (or 2a while ub = ∞)
        const double hl = (lb + a) / 2, hu = hd_isfinite(ub) ? (a + ub) / 2 : 2.0 * a;
        if (int rc = ls_find_feasible_t(ls, a, lb, evals, phi, dphi, flag, ev, hl, hu, (hl + a) / 2, (a + hu) / 2)) return rc;
        if (flag != CGO_FEASIBLE) { o = ls_out(phi0, 0.0, 0, CGO_CANNOT_FIND_FEASIBLE_STEP); return 0; }
    }
    o = ls_out(phi, a, evals, CGO_LINESEARCH_MAX_ITERS_REACHED);
    return 0;
}

// geometric.jl:164-186  evalbacktrackcondition(::Armijo, …)
CGO_HD inline bool armijo_test(double c1, double phi_a, double a, double phi0, double d0) {
    if (!hd_isfinite(phi0) || !hd_isfinite(phi_a) || !hd_isfinite(a)) return false;
    return (phi0 - phi_a) >= -c1 * a * d0;
}

// ---- on-device controller ---------------------------------------------------------------------
constexpr int CTL_MAXP = 7;    // most trial points of a launch (cgo_kernels_cg.hip.hpp: MAXP)
constexpr int CTL_NSUMS = 56;  // widest row: 7 × 7 trial sums + g·u, u·u, padded

struct CtlConfig {
    cgo_ls_config ls;
    double eps, mu;
    int32_t beta_kind, maxp;   // maxp: trial points per launch (1, 3, 5 or 7) = the kernel variant's row layout
    int64_t max_iters;
};

struct CtlState {              // what the NEXT fused launch (accept + dir + trials) consumes
    double f_x, gg;            // objective and g·g at the current iterate (before that launch's accept)
    double a_acc, beta;        // step to accept (= a_initial of the following line search), β of the direction update
    double a[CTL_MAXP];        // trial steps of the following line search
    int32_t npts, go;          // go = 0: launches become no-ops, the host takes over
    int64_t it;                // completed outer iterations
};

struct CtlRecord {             // one per round, published to the host
    double sums[CTL_NSUMS];    // the launch's reduced sums (7 per trial point, then g·u, u·u)
    double a_acc, beta;        // the arguments the launch ran with — the host checks them bit for bit
    double a[CTL_MAXP];
    int32_t npts;              // −1: the round did not run (the controller had already stopped)
    int32_t accepted;          // the controller's line search succeeded inside the launch and armed the next round
    int64_t xwait;             // multi-rank rounds: 100 MHz ticks the finisher spent exchanging blocks with its peers' GPUs
};

// trial results of the finished launch, looked up by step; a miss aborts the search (the host takes over)
struct CtlCacheEval {
    const double *sums; const double *a; int n; int last;
    CGO_HD int operator()(double step, double &phi, double &dphi, double, double, double, double) {
        for (int j = 0; j < n; ++j)
            if (a[j] == step) { phi = sums[7 * j]; dphi = sums[7 * j + 1]; last = j; return 0; }
        return 1;
    }
};
struct CtlNoBackend {          // the bracket-collapse branch of the Wolfe bisection needs vector work: host only
    CGO_HD int is_neg_grad(bool &) { return 2; }
    CGO_HD int reset_dir(double &) { return 2; }
};

// Round logic: the launch just finished applied (a_acc, beta) and evaluated trials at s.a[0..npts); `sums` are its
// global sums.  Run the reference's line search over those points — the SAME state machine the host runs — and,
// iff it returns :success without needing a point the launch did not evaluate and nothing else needs the host
// (non-descent direction, extreme-range norm, non-finite values, stop test, last iteration), advance the state for
// the next launch; otherwise clear `go`.
// ctl_decide: everything but the copy of the sums into the record — on the device 56 lanes do that copy at once
// (one lane alone needs ≈ 3 µs for it: a dependent LDS read + write per word).
CGO_HD inline bool ctl_decide(const CtlConfig &c, CtlState &s, const double *sums, CtlRecord &rec);
CGO_HD inline bool ctl_step(const CtlConfig &c, CtlState &s, const double *sums, CtlRecord &rec) {
    for (int i = 0; i < CTL_NSUMS; ++i) rec.sums[i] = sums[i];
    return ctl_decide(c, s, sums, rec);
}
CGO_HD inline bool ctl_decide(const CtlConfig &c, CtlState &s, const double *sums, CtlRecord &rec) {
    rec.a_acc = s.a_acc; rec.beta = s.beta;
    for (int j = 0; j < CTL_MAXP; ++j) rec.a[j] = s.a[j];
    rec.npts = s.npts; rec.accepted = 0; rec.xwait = 0;
    const int dbase = 7 * c.maxp;  // row layout of the launch (cgo_kernels_cg.hip.hpp)
    const double d0 = sums[dbase];
    double uu = sums[dbase + 1];
    CtlCacheEval ev; ev.sums = sums; ev.a = s.a; ev.n = s.npts; ev.last = -1;
    CtlNoBackend bk;
    LSOut o = ls_out(0.0, 0.0, 0, CGO_INCOMPLETE);
    int rc = 3;
    if (c.ls.kind == CGO_LS_STRONG_WOLFE_BISECTION) rc = ls_strong_wolfe_t(c.ls, s.f_x, d0, s.a_acc, ev, o);
    else if (c.ls.kind == CGO_LS_WOLFE_BISECTION) rc = ls_wolfe_bisection_t(c.ls, s.f_x, d0, uu, s.a_acc, ev, bk, o);
    bool acc = rc == 0 && o.status == CGO_SUCCESS && ev.last >= 0;
    TrialSums t;
    t.f = t.gtu = t.gtgt = t.gtg = t.yy = t.uy = t.ygt = 0.0;
    double a = 0.0, norm = 0.0;
    if (acc) {
        const double *q = sums + 7 * ev.last;   // the accepted trial = the last one the search evaluated
        t.f = q[0]; t.gtu = q[1]; t.gtgt = q[2]; t.gtg = q[3]; t.yy = q[4]; t.uy = q[5]; t.ygt = q[6];
        a = s.a[ev.last];
        if (!(t.gtgt >= 1e-280 && t.gtgt <= 1e300)) acc = false;  // LinearAlgebra.norm rare path → host
        norm = __builtin_sqrt(t.gtgt);
        if (!hd_isfinite(t.f) || !hd_isfinite(norm)) acc = false; // optim.jl:108-121 → host
    }
    if (acc) {
        const int64_t it_new = s.it + 1;
        // optim.jl:53-80,162-169 → the host finishes.  The margin hands every borderline stop test to the
        // host as well, so the two sides can never decide it differently.
        if (it_new >= c.max_iters || !(norm >= c.eps * (1.0 + 1e-9))) acc = false;
    }
    const double a_next = ls_first_step(c.ls, a);                  // optim.jl:92
    if (!hd_isfinite(a_next)) acc = false;
    if (acc && !beta_norms_fast_ok(c.beta_kind, t, uu)) acc = false;   // a norm of getβ needs the scaled rare path → host
    if (!acc) { s.go = 0; return false; }
    const double beta = beta_from_sums(c.beta_kind, c.mu, t, d0, s.gg, uu, beta_norms_fast(t, uu));  // optim.jl:130-135
    rec.accepted = 1;
    s.f_x = t.f; s.gg = t.gtgt; s.it = s.it + 1;                  // optim.jl:136-141
    s.a_acc = a; s.beta = beta;
    double pts[CTL_MAXP];
    if (c.maxp >= 5) {
        s.npts = ls_trial_points_n(c.ls, a_next, c.maxp >= 7 ? 7 : 5, pts);
    } else {
        double p3[3];
        s.npts = ls_trial_points(c.ls, a_next, c.maxp >= 3, p3);
        for (int j = 0; j < 3; ++j) pts[j] = p3[j];
    }
    double lastp = pts[0];   // slots past npts repeat the last point (a launch always evaluates its full row)
    for (int j = 0; j < CTL_MAXP; ++j) {
        if (j < s.npts) lastp = pts[j];
        s.a[j] = lastp;
    }
    return true;
}

}  // namespace cgo
