// cgo_resident.hpp — whole outer iterations WITHOUT the host: the scalar side of the resident solver.
//
// For cache-sized problems (BASELINE configs 1 and 2: n = 1e3 … ≈ 1.5e6) a launch per trial is the wrong unit: the vector
// work of a trial is 1–15 µs and everything around it — kernel boundary, publish over PCIe, host decision, launch — costs
// as much again (config 1 ran at 17 k it/s on the GPU against 44–80 k on one CPU thread).  The resident
// solver (k_resident, cgo_kernels_resident.hip.hpp) keeps x, u and the objective's parameter vector in the LDS of up to
// 256 workgroups for a whole slice of outer iterations; every thread of every workgroup runs THIS loop — the outer loop of
// optim.jl:50-160 with the reference's line searches (cgo_ctl.hpp: ls_strong_wolfe_t, ls_wolfe_bisection_t, the same
// templates the host engine runs) — on identical global sums, so all of them take identical decisions: replicated
// control flow inside the chip, exactly as the ranks of a sharded solve replicate it across GPUs.
//
// `res_iterate` is a template over the vector operations (`V`): on the device they are LDS passes + an all-gather of one
// row per workgroup; in tests/hostsim they are plain loops — the CPU tier runs this very loop against the oracle.
//
// What the loop does NOT do it hands back (RES_HOST) BEFORE touching x or u, and the host engine runs that one iteration
// as it always has: every line-search outcome other than :success, non-finite values, the rare-path norms of
// LinearAlgebra.norm, the bracket-collapse branch of WolfeBisection (vector work), Backtracking, quasi-Newton flavours.
#pragma once

#include "cgo_ctl.hpp"

namespace cgo {

constexpr int RES_MAXP = 7;
// On the device EVERYTHING of the loop is inlined into the kernel: a call boundary would force the loop state (ResState,
// the trial cache, the points) out of registers into scratch memory — measured: 47 µs per outer iteration at n = 1000
// with the evaluator as a function of its own, against single digits with the state in registers (round 3).
#if defined(__HIP_DEVICE_COMPILE__)
#define RES_EV_INLINE __attribute__((always_inline))
#else
#define RES_EV_INLINE
#endif

struct ResConfig {
    cgo_ls_config ls;
    double eps, mu;
    int32_t beta_kind, npts;   // npts: trial steps per pass (1, 3 or 7): the requested step + speculative candidates
    int64_t max_iters;
    int32_t log_on, pad_;
};

enum ResReason : int32_t {
    RES_BUDGET = 0,    // the slice is done
    RES_STOP = 1,      // the stop test at the top of an iteration fired (optim.jl:53-80,162-169): the host finishes
    RES_HOST = 2,      // the iteration at s.it + 1 needs the host; x, u are those of iteration s.it
    RES_LOG_FULL = 3,  // the trial log must be drained first
    RES_ERROR = 4      // the exchange between workgroups gave up (state unusable)
};

struct ResState {
    // the loop state of optim.jl:25-47,136-145 between two outer iterations
    double f_x, gg, norm, dphi0, uu, a_initial;
    int64_t it;                // completed outer iterations
    int64_t evals;             // evalϕdϕ! calls of the iterations completed in this slice
    int32_t ncache, dir_neg;   // trial results already known for the next line search (the fused pass of the last accept)
    double ca[RES_MAXP];
    TrialSums cs[RES_MAXP];
    double last_a, last_beta;
    // outcome of the slice
    int64_t done, log_len, passes;
    int32_t reason, pad_;
    int64_t t_total, t_compute, t_reduce, t_exchange;   // device only: 100 MHz ticks of the slice and of its passes' three phases
    int64_t t_machine, t_eval, t_post;                  // … of the line-search machine's steps, of evalϕdϕ! (passes included), of getβ & co.
    int64_t t_cycles;                                   // shader clock cycles of the slice (s_memtime): ÷ t_total = the clock the CU ran at
};

struct ResRecord { double f, norm, a, beta; int64_t evals; };   // one per completed iteration → the trace (types.jl:56-79)
struct ResLog { double a, phi, dphi; };

constexpr int64_t RES_LOG_MARGIN = 2048;   // a slice stops for a drain when fewer free log entries than this remain

CGO_HD inline bool res_same_bits(double a, double b) {
    unsigned long long x, y;
    __builtin_memcpy(&x, &a, 8); __builtin_memcpy(&y, &b, 8);
    return x == y;
}

// ---- the two bisection line searches as RESUMABLE machines ---------------------------------------------------------------
// cgo_ctl.hpp states them as loops over an evaluator callback (ls_strong_wolfe_t / ls_zoom_t, ls_wolfe_bisection_t /
// ls_find_feasible_t): right for the host, whose evaluator launches a kernel.  Inlined into a GPU kernel the callback form
// puts a copy of the whole vector pass at each of its dozen call sites: 30 000 instructions (240 KB of code against a
// 64 KB instruction cache), 4 400 SGPR-spill reloads, ≈ 3 µs of overhead around every evaluation (measured, round 3).
// Here the SAME statements, in the same order, run as a resumable machine: wherever the template calls ev(a, …) the machine
// leaves the step and its hints in `m`, returns 1, and is re-entered with ϕ(a), dϕ(a) — so the caller evaluates at ONE
// place.  Returns 0: finished (m.o); 1: evaluate m.a (hints m.h1…m.h4); ≥ 2: aborted (the bracket-collapse branch of
// wolfe.jl:122-133 needs vector work: the host's).  tests/test_hostsim.py holds the machines bitwise to the templates
// (every parity, status and reset case: same steps, same hints → same launches, same results).
struct LsMachine {
    int32_t state, flag;
    double phi0, d0, uu, a_initial;          // inputs
    double a, h1, h2, h3, h4;                // the evaluation asked for
    LSOut o;
    double phi, dphi; int64_t evals;
    double a_prev, phi_prev; int64_t k;      // strong Wolfe (nocedal.jl:33-158)
    double lo, hi, phi_lo; int64_t kz; int32_t run_hi, run_lo;   // zoom! (nocedal.jl:162-209)
    double a_first, lb, ub, ff_lb; int64_t iter;                 // Wolfe bisection + findfeasiblestepsize! (wolfe.jl:13-207)
};

// Written as FLAT transition functions — one switch on the state, straight-line code per transition, a small `pc` loop where a
// transition can chain into the next without an evaluation — not as a protothread (switch into the middle of the template's
// loops): that form is irreducible control flow, which the GPU compiler repairs with a dispatch chain of a dozen dependent
// branches per re-entry (measured: ≈ 0.4 µs per step of the search at 2.4 GHz, a quarter of an outer iteration at n = 1000).

// nocedal.jl:33-209 — ls_strong_wolfe_t + ls_zoom_t (cgo_ctl.hpp), the same statements in the same order
enum : int32_t { SW_START = 0, SW_MAIN = 1, SW_ZOOM = 2 };
CGO_HD inline int ls_sw_machine(const cgo_ls_config &ls, LsMachine &m, double phi_in, double dphi_in) {
    const double c1 = ls.c1, c2 = ls.c2, phi0 = m.phi0, d0 = m.d0, growth = ls.a_max_growth_factor;
    // what the head of each loop does before its ev(): the step, the hints, the request
    auto main_head = [&]() -> int {           // for (k …) { hz, he; ev(a, …)
        if (!(m.k < ls.max_iters)) { m.o = ls_out(m.phi, m.a, m.evals, CGO_LINESEARCH_MAX_ITERS_REACHED); return 0; }
        const double hz = (m.a_prev + m.a) / 2, he = (m.a * growth + m.a) / 2;
        if (m.k >= 3) { m.h1 = he; m.h2 = (he * growth + he) / 2; m.h3 = hz; m.h4 = (m.a + he) / 2; }
        else { m.h1 = hz; m.h2 = he; m.h3 = (hz + m.a) / 2; m.h4 = (m.a + he) / 2; }
        m.state = SW_MAIN;
        return 1;
    };
    auto zoom_head = [&]() -> int {           // for (kz …) { a = (lo + hi)/2; hl, hu; ev(a, …)
        if (!(m.kz < ls.zoom_max_iters)) { m.o = ls_out(m.phi, m.a, m.evals, CGO_ZOOM_MAX_ITERS_REACHED); return 0; }
        m.a = (m.lo + m.hi) / 2;
        const double hl = (m.lo + m.a) / 2, hu = (m.a + m.hi) / 2;
        if (m.run_hi >= 2) { m.h1 = hl; m.h2 = (m.lo + hl) / 2; m.h3 = hu; m.h4 = (hl + m.a) / 2; }
        else if (m.run_lo >= 2) { m.h1 = hu; m.h2 = (hu + m.hi) / 2; m.h3 = hl; m.h4 = (m.a + hu) / 2; }
        else { m.h1 = hl; m.h2 = hu; m.h3 = (hl + m.a) / 2; m.h4 = (m.a + hu) / 2; }
        m.state = SW_ZOOM;
        return 1;
    };
    auto zoom_enter = [&](double lo, double hi, double phi_lo) -> int {   // zoom!(…, lo, hi, ϕ_lo, …)
        m.lo = lo; m.hi = hi; m.phi_lo = phi_lo;
        m.a = 0; m.phi = 0; m.dphi = 0; m.run_hi = 0; m.run_lo = 0; m.kz = 0;
        return zoom_head();
    };
    switch (m.state) {
    case SW_START:
        m.a = ls_first_step(ls, m.a_initial);
        if (d0 > 0.0) { m.o = ls_out(phi0, 0.0, 0, CGO_NON_DESCENT_SEARCH_DIRECTION); return 0; }
        m.a_prev = 0.0; m.phi_prev = phi0; m.phi = phi0; m.dphi = d0; m.evals = 0; m.k = 0;
        return main_head();
    case SW_MAIN: {
        m.phi = phi_in; m.dphi = dphi_in;
        ++m.evals;
        const bool too_high = m.phi > phi0 + c1 * m.a * d0;
        const bool not_lower = m.phi >= m.phi_prev;
        if (too_high || (not_lower && m.k > 0)) return zoom_enter(m.a_prev, m.a, m.phi_prev);
        if (__builtin_fabs(m.dphi) <= -c2 * d0) { m.o = ls_out(m.phi, m.a, m.evals, CGO_SUCCESS); return 0; }
        if (m.dphi >= 0) return zoom_enter(m.a, m.a_prev, m.phi);
        m.a_prev = m.a;
        m.phi_prev = m.phi;
        const double a_max = m.a * growth;
        if (m.a > a_max) { m.o = ls_out(m.phi, m.a, m.evals, CGO_LINESEARCH_A_MAX_OVERFLOW); return 0; }
        m.a = (a_max + m.a) / 2;
        ++m.k;
        return main_head();
    }
    case SW_ZOOM:
        m.phi = phi_in; m.dphi = dphi_in;
        ++m.evals;
        if ((m.phi > phi0 + c1 * m.a * d0) || (m.phi >= m.phi_lo)) {
            m.hi = m.a;
            ++m.run_hi; m.run_lo = 0;
        } else {
            if (__builtin_fabs(m.dphi) <= -c2 * d0) { m.o = ls_out(m.phi, m.a, m.evals, CGO_SUCCESS); return 0; }
            if (m.dphi * (m.hi - m.lo) >= 0) { m.hi = m.lo; m.run_lo = 0; } else ++m.run_lo;
            m.run_hi = 0;
            m.lo = m.a;
            m.phi_lo = m.phi;
        }
        ++m.kz;
        return zoom_head();
    }
    return 3;
}

// wolfe.jl:13-207 — ls_wolfe_bisection_t with ls_find_feasible_t inlined at its two call sites (m.flag's upper half says
// which: 0 the initial one, 1 the one inside the loop); the bracket collapse aborts (2)
enum : int32_t { WB_START = 0, WB_FF_FIRST = 1, WB_FF_LOOP = 2 };
enum : int32_t { WB_PC_FF_BEGIN = 0, WB_PC_FF_CHECK = 1, WB_PC_FF_DONE = 2, WB_PC_ITER = 3 };
CGO_HD inline int ls_wb_machine(const cgo_ls_config &ls, LsMachine &m, double phi_in, double dphi_in) {
    const double phi0 = m.phi0, d0 = m.d0, inf = __builtin_inf();
    int pc;
    switch (m.state) {
    case WB_START:
        m.a_first = ls_first_step(ls, m.a_initial);
        if (!hd_isfinite(phi0)) { m.o = ls_out(phi0, 0.0, 0, CGO_ACCEPTED_NON_FINITE_ITERATE); return 0; }
        if (d0 > 0.0) { m.o = ls_out(phi0, 0.0, 0, CGO_NON_DESCENT_SEARCH_DIRECTION); return 0; }
        m.a = m.a_first; m.lb = 0.0; m.ub = inf; m.phi = 0; m.dphi = 0; m.evals = 0; m.flag = 0; m.k = 0;
        m.ff_lb = 0.0; m.run_lo = 0;   // run_lo: which call site of findfeasiblestepsize! is running (0 initial, 1 in the loop)
        m.h1 = (m.lb + m.a) / 2; m.h2 = 2.0 * m.a; m.h3 = ((m.lb + m.a) / 2 + m.a) / 2; m.h4 = (m.a + 2.0 * m.a) / 2;
        pc = WB_PC_FF_BEGIN;
        break;
    case WB_FF_FIRST:        // the first ev() of findfeasiblestepsize! returned
        m.phi = phi_in; m.dphi = dphi_in; ++m.evals;
        m.iter = 1;
        pc = WB_PC_FF_CHECK;
        break;
    case WB_FF_LOOP:         // an ev() inside its halving loop returned
        m.phi = phi_in; m.dphi = dphi_in; ++m.evals;
        ++m.iter;
        pc = WB_PC_FF_CHECK;
        break;
    default:
        return 3;
    }
    for (;;) {
        if (pc == WB_PC_FF_BEGIN) {            // findfeasiblestepsize!(…, a, lb = ff_lb, …): wolfe.jl:171-207
            if (m.ff_lb > m.a) { m.phi = 0.0; m.dphi = 0.0; m.flag = CGO_BISECTION_LOWER_BOUND_LARGER_THAN_PROPOSED_STEP; pc = WB_PC_FF_DONE; continue; }
            m.state = WB_FF_FIRST;
            return 1;
        }
        if (pc == WB_PC_FF_CHECK) {            // for (iter = 1; a > lb && iter < feasibility_max_iters; ++iter) { …
            if (m.a > m.ff_lb && m.iter < ls.feasibility_max_iters) {
                if (hd_isfinite(m.phi) && hd_isfinite(m.dphi)) { m.flag = CGO_FEASIBLE; pc = WB_PC_FF_DONE; continue; }
                m.a = m.a * 0.5;
                m.h1 = m.h2 = m.h3 = m.h4 = __builtin_nan("");
                m.state = WB_FF_LOOP;
                return 1;
            }
            m.flag = CGO_INFEASIBLE;
            pc = WB_PC_FF_DONE;
            continue;
        }
        if (pc == WB_PC_FF_DONE) {
            if (m.run_lo == 0) {               // the initial call (wolfe.jl:51-60)
                if (m.flag != CGO_FEASIBLE) { m.o = ls_out(phi0, 0.0, 0, CGO_CANNOT_FIND_INITIAL_FEASIBLE_STEP); return 0; }
                m.k = 0;
            } else {                           // the call inside the loop (wolfe.jl:136-150)
                if (m.flag != CGO_FEASIBLE) { m.o = ls_out(phi0, 0.0, 0, CGO_CANNOT_FIND_FEASIBLE_STEP); return 0; }
                ++m.k;
            }
            pc = WB_PC_ITER;
            continue;
        }
        // WB_PC_ITER: for (k …) { …
        if (!(m.k < ls.max_iters)) { m.o = ls_out(m.phi, m.a, m.evals, CGO_LINESEARCH_MAX_ITERS_REACHED); return 0; }
        bool ok_large, ok_small;
        wolfe_tests(ls, phi0, d0, m.uu, m.phi, m.dphi, m.a, ok_large, ok_small);
        if (ok_large && ok_small) { m.o = ls_out(m.phi, m.a, m.evals, CGO_SUCCESS); return 0; }
        if (!ok_large) {
            m.ub = m.a;
            m.a = (m.lb + m.ub) / 2;
        } else {
            m.lb = m.a;
            if (!hd_isfinite(m.ub)) {
                m.a = 2.0 * m.a;
                if (m.a > ls.max_step_size) { m.o = ls_out(phi0, 0.0, 0, CGO_MAX_STEP_LENGTH_REACHED); return 0; }
            } else {
                m.a = (m.lb + m.ub) / 2;
            }
        }
        if (!(m.lb < m.a && m.a < m.ub)) return 2;   // bracket collapsed (wolfe.jl:122-133): ‖u + g‖ and u ← −g are vector work
        const double hl = (m.lb + m.a) / 2, hu = hd_isfinite(m.ub) ? (m.a + m.ub) / 2 : 2.0 * m.a;
        m.h1 = hl; m.h2 = hu; m.h3 = (hl + m.a) / 2; m.h4 = (m.a + hu) / 2;
        m.ff_lb = m.lb;
        m.run_lo = 1;
        pc = WB_PC_FF_BEGIN;
    }
}

// evalϕdϕ! for the line-search templates: a result the last pass already produced, or one new pass that evaluates `a`
// together with the hinted candidate steps (the mirror of Solver::evaln, cgo_engine.cpp).
template <class V>
struct ResEval {
    const ResConfig &c; ResState &s; V &v;
    ResLog *log; int64_t log_cap; int64_t log_len;   // entries of the current iteration go to log[log_len …)
    int64_t evals = 0;
    TrialSums last; double last_a;
    bool overflow = false;
    // (not inlined: the line-search templates call it from a dozen places, and on the device each copy would carry a
    //  whole pass over the chunk with its accumulators.  Every array index below is a compile-time constant after
    //  unrolling, so that cache and points live in registers, not in scratch memory.)
    CGO_HD RES_EV_INLINE int operator()(double a, double &phi, double &dphi, double h1, double h2, double h3, double h4) {
        constexpr int NC = V::kNpts > 0 ? V::kNpts : RES_MAXP;   // cache entries a pass of this width can leave (fewer moves and selects)
        bool hit = false;
        TrialSums found = s.cs[0];
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (!hit && j < s.ncache && res_same_bits(a, s.ca[j])) { hit = true; found = s.cs[j]; }
        if (!hit) {
            // requested step + up to two distinct, finite, positive hints (a trial-only pass is almost always the last
            // of its line search: no grandchildren) — the mirror of Solver::evaln
            double p0 = a, p1 = a, p2 = a;
            int k = 1;
            const int np_ = V::kNpts > 0 ? V::kNpts : c.npts;
            const int mp = np_ < 3 ? np_ : 3;
            const double hs[4] = {h1, h2, h3, h4};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double h = hs[q];
                const bool ok = k < mp && hd_isfinite(h) && h > 0.0 && h != p0 && (k < 2 || h != p1);
                if (ok) { if (k == 1) p1 = h; else p2 = h; ++k; }
            }
            if (k == 1) { p1 = a; p2 = a; } else if (k == 2) p2 = p1;   // padding: a repeated point costs nothing but its flops
            const double pts[3] = {p0, p1, p2};
            TrialSums out[3];
            if (int rc = v.trial(pts, k, out)) return rc;
            s.passes++;
            s.ncache = k;
#pragma unroll
            for (int j = 0; j < 3; ++j) { s.ca[j] = pts[j]; s.cs[j] = out[j]; }
            found = out[0];
        }
        last = found; last_a = a;
        ++evals;
        phi = last.f; dphi = last.gtu;
        if (c.log_on) {
            if (log_len < log_cap) {
                if (v.leader()) { log[log_len].a = a; log[log_len].phi = phi; log[log_len].dphi = dphi; }
                ++log_len;
            } else overflow = true;
        }
        return 0;
    }
};

// ls_trial_points / ls_trial_points_n (cgo_ctl.hpp) with compile-time array indices only (the device keeps pts[] in
// registers): the requested step, then its distinct, finite, positive candidates in the same order — both candidates of
// the first decision (maxp ≥ 3), then the grandchildren / great-grandchildren towards a0 (maxp ≥ 5 / 7).  Unused entries 0.
CGO_HD inline int res_trial_points(const cgo_ls_config &ls, double a0, int maxp, double (&pts)[RES_MAXP]) {
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
    const int ncand = maxp >= 5 ? 6 : (maxp >= 3 ? 2 : 0);
#pragma unroll
    for (int j = 0; j < RES_MAXP; ++j) pts[j] = 0.0;
    pts[0] = a0;
    int k = 1;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const double v = cand[q];
        bool ok = q < ncand && k < maxp && hd_isfinite(v) && v > 0.0;
#pragma unroll
        for (int j = 0; j < RES_MAXP; ++j) if (j < k && v == pts[j]) ok = false;
        if (ok) {
#pragma unroll
            for (int j = 1; j < RES_MAXP; ++j) if (j == k) pts[j] = v;
            ++k;
        }
    }
    return k;
}

// Up to `budget` outer iterations of minimizeobjective (optim.jl:50-160).  Every caller thread passes identical arguments
// and V returns identical sums to all of them.  recs[0 … s.done) and log[0 … s.log_len) are written by V's leader only.
template <class V>
CGO_HD inline void res_iterate(const ResConfig &c, ResState &s, V &v, int64_t budget, ResRecord *recs, ResLog *log, int64_t log_cap) {
    s.done = 0; s.log_len = 0; s.evals = 0; s.passes = 0; s.reason = RES_BUDGET;
    s.t_machine = 0; s.t_eval = 0; s.t_post = 0;
    const int npts = V::kNpts > 0 ? V::kNpts : c.npts;   // a compile-time constant on the device
    constexpr int NCP = V::kNpts > 0 ? V::kNpts : RES_MAXP;
    while (s.done < budget) {
        const int64_t n = s.it + 1;
        if (n > c.max_iters) { s.reason = RES_STOP; break; }                                          // optim.jl:162-169
        if (hd_isfinite(s.f_x) && hd_isfinite(s.norm) && s.norm < c.eps) { s.reason = RES_STOP; break; }   // optim.jl:53-80
        if (c.log_on && s.log_len + RES_LOG_MARGIN > log_cap) { s.reason = RES_LOG_FULL; break; }
        ResEval<V> ev{c, s, v, log, log_cap, s.log_len};
        LsMachine m{};
        m.state = 0; m.flag = 0; m.phi0 = s.f_x; m.d0 = s.dphi0; m.uu = s.uu; m.a_initial = s.a_initial;
        m.o = ls_out(0.0, 0.0, 0, CGO_INCOMPLETE);
        const bool strong = c.ls.kind == CGO_LS_STRONG_WOLFE_BISECTION;
        int rc = 3;
        {
            double phi = 0.0, dphi = 0.0;
            for (;;) {   // the line search asks, ONE place evaluates
                const long long tm0 = v.clock();
                rc = strong ? ls_sw_machine(c.ls, m, phi, dphi) : ls_wb_machine(c.ls, m, phi, dphi);
                const long long tm1 = v.clock();
                s.t_machine += tm1 - tm0;
                if (rc != 1) break;
                const int erc = ev(m.a, phi, dphi, m.h1, m.h2, m.h3, m.h4);
                s.t_eval += v.clock() - tm1;
                if (erc) { rc = erc; break; }
            }
        }
        const LSOut o = m.o;
        const long long tp0 = v.clock();
        if (rc == 9) { s.reason = RES_ERROR; break; }
        if (rc != 0 || o.status != CGO_SUCCESS || ev.evals == 0 || ev.overflow) { s.reason = RES_HOST; break; }   // optim.jl:93-104 → host
        const TrialSums t = ev.last;            // info.xp / df_xp: the LAST evaluated trial = the accepted one (both bisection searches)
        if (!sumsq_in_range(t.gtgt)) { s.reason = RES_HOST; break; }                                   // LinearAlgebra.norm rare path
        const double norm_xp = __builtin_sqrt(t.gtgt);                                                // optim.jl:107
        if (!hd_isfinite(o.phi) || !hd_isfinite(norm_xp)) { s.reason = RES_HOST; break; }              // optim.jl:108-121
        if (!beta_norms_fast_ok(c.beta_kind, t, s.uu)) { s.reason = RES_HOST; break; }
        BetaNorms bn = beta_norms_fast(t, s.uu);
        bn.gt = norm_xp;
        const double beta = beta_from_sums(c.beta_kind, c.mu, t, s.dphi0, s.gg, s.uu, bn);            // optim.jl:130-135
        // optim.jl:136-141,152-159
        s.f_x = o.phi; s.norm = norm_xp; s.gg = t.gtgt; s.it = n; s.a_initial = o.a;
        s.evals += ev.evals; s.log_len = ev.log_len;
        s.last_a = ev.last_a; s.last_beta = beta;
        if (v.leader()) { ResRecord &r = recs[s.done]; r.f = s.f_x; r.norm = s.norm; r.a = o.a; r.beta = beta; r.evals = o.evals; }
        s.done++;
        const bool will_stop = (n == c.max_iters) || (hd_isfinite(s.f_x) && hd_isfinite(s.norm) && s.norm < c.eps);
        const double a_next = ls_first_step(c.ls, s.a_initial);                                       // optim.jl:92
        double pts[RES_MAXP] = {0, 0, 0, 0, 0, 0, 0};
        int k = 0;
        if (!will_stop && hd_isfinite(a_next)) k = res_trial_points(c.ls, a_next, npts >= 7 ? 7 : (npts >= 5 ? 5 : (npts >= 3 ? 3 : 1)), pts);
        // x ← xp, updatedir! (optim.jl:136-145) and the first trials of the next line search in ONE pass
        {   // pad: a pass evaluates a fixed number of points; the spare ones repeat the last real step (results ignored)
            double lastp = 0.0;
#pragma unroll
            for (int j = 0; j < NCP; ++j) { if (j < k) lastp = pts[j]; else pts[j] = lastp; }
        }
        TrialSums out[RES_MAXP] = {};
        double gu = 0.0, uu_new = 0.0;
        s.t_post += v.clock() - tp0;
        if (int rc2 = v.accept_dir_trial(ev.last_a, beta, pts, k, out, gu, uu_new)) { (void)rc2; s.reason = RES_ERROR; break; }
        s.passes++;
        s.dphi0 = gu; s.uu = uu_new; s.dir_neg = (beta == 0.0) ? 1 : 0;
        s.ncache = k;
#pragma unroll
        for (int j = 0; j < NCP; ++j) { s.ca[j] = pts[j]; s.cs[j] = out[j]; }   // (entries ≥ k are never looked at)
    }
}

}  // namespace cgo
