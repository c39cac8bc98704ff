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
};

struct ResRecord { double f, norm, a, beta; int64_t evals; };   // one per completed iteration → the trace (types.jl:56-79)
struct ResLog { double a, phi, dphi; };

constexpr int64_t RES_LOG_MARGIN = 2048;   // a slice stops for a drain when fewer free log entries than this remain

CGO_HD inline bool res_same_bits(double a, double b) {
    unsigned long long x, y;
    __builtin_memcpy(&x, &a, 8); __builtin_memcpy(&y, &b, 8);
    return x == y;
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
    CGO_HD int operator()(double a, double &phi, double &dphi, double h1, double h2, double h3, double h4) {
        int hit = -1;
        for (int j = 0; j < s.ncache; ++j)
            if (res_same_bits(a, s.ca[j])) { hit = j; break; }
        if (hit < 0) {
            double pts[RES_MAXP] = {a, 0, 0, 0, 0, 0, 0};
            const double hs[4] = {h1, h2, h3, h4};
            int k = 1;
            const int mp = c.npts < 3 ? c.npts : 3;   // a trial-only pass is almost always the last of its line search
            for (int q = 0; q < 4 && k < mp; ++q) {
                const double h = hs[q];
                bool ok = hd_isfinite(h) && h > 0.0;
                for (int j = 0; ok && j < k; ++j) ok = (h != pts[j]);
                if (ok) pts[k++] = h;
            }
            TrialSums out[RES_MAXP];
            if (int rc = v.trial(pts, k, out)) return rc;
            s.passes++;
            s.ncache = k;
            for (int j = 0; j < k; ++j) { s.ca[j] = pts[j]; s.cs[j] = out[j]; }
            hit = 0;
        }
        last = s.cs[hit]; last_a = a;
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

// Up to `budget` outer iterations of minimizeobjective (optim.jl:50-160).  Every caller thread passes identical arguments
// and V returns identical sums to all of them.  recs[0 … s.done) and log[0 … s.log_len) are written by V's leader only.
template <class V>
CGO_HD inline void res_iterate(const ResConfig &c, ResState &s, V &v, int64_t budget, ResRecord *recs, ResLog *log, int64_t log_cap) {
    s.done = 0; s.log_len = 0; s.evals = 0; s.passes = 0; s.reason = RES_BUDGET;
    while (s.done < budget) {
        const int64_t n = s.it + 1;
        if (n > c.max_iters) { s.reason = RES_STOP; break; }                                          // optim.jl:162-169
        if (hd_isfinite(s.f_x) && hd_isfinite(s.norm) && s.norm < c.eps) { s.reason = RES_STOP; break; }   // optim.jl:53-80
        if (c.log_on && s.log_len + RES_LOG_MARGIN > log_cap) { s.reason = RES_LOG_FULL; break; }
        ResEval<V> ev{c, s, v, log, log_cap, s.log_len};
        LSOut o = ls_out(0.0, 0.0, 0, CGO_INCOMPLETE);
        int rc = 3;
        double uu = s.uu;
        if (c.ls.kind == CGO_LS_STRONG_WOLFE_BISECTION) rc = ls_strong_wolfe_t(c.ls, s.f_x, s.dphi0, s.a_initial, ev, o);
        else if (c.ls.kind == CGO_LS_WOLFE_BISECTION) { CtlNoBackend bk; rc = ls_wolfe_bisection_t(c.ls, s.f_x, s.dphi0, uu, s.a_initial, ev, bk, o); }
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
        if (!will_stop && hd_isfinite(a_next)) {
            if (c.npts >= 5) k = ls_trial_points_n(c.ls, a_next, c.npts >= 7 ? 7 : 5, pts);
            else { double p3[3]; k = ls_trial_points(c.ls, a_next, c.npts >= 3, p3); for (int j = 0; j < 3; ++j) pts[j] = p3[j]; }
        }
        // x ← xp, updatedir! (optim.jl:136-145) and the first trials of the next line search in ONE pass
        TrialSums out[RES_MAXP];
        double gu = 0.0, uu_new = 0.0;
        if (int rc2 = v.accept_dir_trial(ev.last_a, beta, pts, k, out, gu, uu_new)) { (void)rc2; s.reason = RES_ERROR; break; }
        s.passes++;
        s.dphi0 = gu; s.uu = uu_new; s.dir_neg = (beta == 0.0) ? 1 : 0;
        s.ncache = k;
        for (int j = 0; j < k; ++j) { s.ca[j] = pts[j]; s.cs[j] = out[j]; }
    }
}

}  // namespace cgo
