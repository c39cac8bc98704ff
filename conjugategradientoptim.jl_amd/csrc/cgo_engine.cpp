// cgo_engine.cpp — scalar control plane of the iteration engine.
//
// What the reference does with six n-vectors and ~28+7k vector passes per outer
// iteration is re-expressed here as a state machine over the
// handful of reduced scalars that the fused device launches return:
//
//   steady state, first trial accepted:  ONE launch per outer iteration
//     KK_ACCEPT_DIR_TRIAL = { x ← xp ; g ← g⁺ ; u ← −g + βu ; dϕ₀,u·u ;
//                             xp' = x + a'u ; g⁺' = ∇f(xp') ; ϕ,dϕ,β-partials }
//   further trials of a line search: a launch evaluates the requested step together with the
//   steps the search can ask for next (3, 5 or 7 per launch: cgo_ctl.hpp, ls_trial_points_n), so
//   most line searches finish inside the launch that started them.
//
// The first trial of the next line search is launched speculatively together
// with the direction update because its step (the previous a*, optim.jl:92, or
// the sanitised default, nocedal.jl:49-52 / wolfe.jl:30-32) is known before the
// line search starts.  If the line search then exits before evaluating
// (non-descent direction, nocedal.jl:57-63) the speculative result is dropped.
//
// Build with -ffp-contract=off: Julia evaluates a*b+c unfused and the branch
// tests below must round like the reference's.
#include "cgo_engine.hpp"

#include <algorithm>
#include <cstring>

namespace cgo {

static const char *kStatusNames[CGO_NUM_STATUS] = {
    "incomplete",
    "success",
    "increasing_objective",
    "non_finite_objective_or_gradient_proposed",
    "max_iters_reached",
    "non_descent_search_direction",
    "linesearch_a_max_overflow",
    "linesearch_max_iters_reached",
    "zoom_max_iters_reached",
    "accepted_non_finite_iterate",
    "cannot_find_initial_feasible_step",
    "max_step_length_reached",
    "cannot_find_feasible_step",
    "step_bracket_precision_issue",
    "bisection_lower_bound_larger_than_proposed_step",
    "feasible",
    "infeasible",
    "non_finite_step_proposed",
    "proposed_step_same_as_current_step",
    "linesearch_failed",
};

const char *status_name(int s) {
    return (s >= 0 && s < CGO_NUM_STATUS) ? kStatusNames[s] : "unknown";
}

static const char *kKernelNames[KK_COUNT] = {
    "init", "trial", "accept_dir_trial", "accept_dir", "accept_only",
    "reset_dir", "upg_norm", "lbfgs_push", "lbfgs_loop", "lbfgs_final", "lse_stats", "lse_grad",
    "scaled_norm", "dir_trial", "sys_project", "resident",
};

const char *kernel_kind_name(int k) { return (k >= 0 && k < KK_COUNT) ? kKernelNames[k] : "unknown"; }

int check_cg_config(const cgo_cg_config *c, std::string &why) {
    if (!c) { why = "null CGConfig"; return CGO_EINVAL; }
    if (!(0.0 < c->eps && c->eps < 1.0)) {  // types.jl:187
        why = "AssertionError: zero(T) < ϵ < one(T)  (types.jl:187)";
        return CGO_EINVAL;
    }
    if (c->beta.kind < 0 || c->beta.kind > CGO_BETA_BROYDEN_FAMILY) { why = "unknown β_config kind"; return CGO_EINVAL; }
    if (c->beta.kind == CGO_BETA_BROYDEN_FAMILY && !(0.0 <= c->beta.mu)) {   // θ travels in the mu field
        why = "AssertionError: zero(T) <= θ  (qn_flavours.jl:57)"; return CGO_EINVAL;
    }
    if (c->beta.kind == CGO_BETA_LBFGS && (c->beta.lbfgs_m < 1 || c->beta.lbfgs_m > 64)) {
        why = "LBFGS history length m must be in 1..64"; return CGO_EINVAL;
    }
    if (c->max_iters < 0) { why = "max_iters must be ≥ 0"; return CGO_EINVAL; }
    return CGO_OK;
}

int check_ls_config(const cgo_ls_config *l, std::string &why) {
    if (!l) { why = "null LineSearchConfig"; return CGO_EINVAL; }
    if (l->kind == CGO_LS_STRONG_WOLFE_BISECTION) {
        if (!(0.0 < l->c1 && l->c1 < l->c2 && l->c2 < 1.0)) {
            why = "AssertionError: zero(T) < c1 < c2 < one(T)  (nocedal.jl:22)"; return CGO_EINVAL;
        }
        if (!(l->max_iters >= 0)) { why = "AssertionError: max_iters >= 0  (nocedal.jl:24)"; return CGO_EINVAL; }
        if (!(l->zoom_max_iters >= 0)) { why = "AssertionError: zoom_max_iters >= 0  (nocedal.jl:25)"; return CGO_EINVAL; }
        if (!(l->a_max_growth_factor > 1.0)) { why = "AssertionError: a_max_growth_factor > 1  (nocedal.jl:26)"; return CGO_EINVAL; }
        return CGO_OK;
    }
    if (l->kind == CGO_LS_WOLFE_BISECTION) {
        if (l->cond_kind == CGO_COND_WOLFE) {
            if (!(0.0 < l->c1 && l->c1 < l->c2 && l->c2 < 1.0)) {
                why = "AssertionError: zero(T) < c1 < c2 < one(T)  (wolfe.jl:278)"; return CGO_EINVAL;
            }
        } else if (l->cond_kind == CGO_COND_YUAN_WEI_LU) {
            if (!(0.0 < l->delta1 && l->delta1 < l->c1 && l->c1 < l->c2 && l->c2 < 1.0)) {
                why = "AssertionError: zero(T) < δ1 < c1 < c2 < one(T)  (wolfe.jl:233)"; return CGO_EINVAL;
            }
        } else { why = "unknown Wolfe condition kind"; return CGO_EINVAL; }
        return CGO_OK;
    }
    if (l->kind == CGO_LS_BACKTRACKING) {
        if (l->cond_kind != CGO_COND_ARMIJO) { why = "Backtracking needs the Armijo condition"; return CGO_EINVAL; }
        if (!(0.0 < l->c1 && l->c1 < 1.0)) {
            why = "AssertionError: zero(T) < c1 < one(T)  (geometric.jl:169)"; return CGO_EINVAL;
        }
        if (!(l->discount_factor > 0.0 && std::isfinite(l->discount_factor))) { why = "discount_factor must be positive"; return CGO_EINVAL; }
        return CGO_OK;
    }
    why = "unknown LineSearchConfig kind";
    return CGO_EINVAL;
}

static inline TrialSums trial_sums(const Scal &t) { return TrialSums{t.f, t.gtu, t.gtgt, t.gtg, t.yy, t.uy, t.ygt}; }

int check_lss_config(const cgo_lss_config *l, std::string &why) {
    if (!l) { why = "null LinesearchSolveSys"; return CGO_EINVAL; }
    if (!(0.0 < l->rho && l->rho < 1.0)) { why = "AssertionError: zero(T) < ρ < one(T)  (solve_system.jl:21)"; return CGO_EINVAL; }
    if (!(l->s > 0.0)) { why = "AssertionError: s > zero(T)  (solve_system.jl:23)"; return CGO_EINVAL; }
    if (l->max_iters < 0) { why = "max_iters must be ≥ 0"; return CGO_EINVAL; }
    return CGO_OK;
}

// round(Int, log(ρ, 1e-6))  (solve_system.jl:17); Julia rounds ties to even
int64_t lss_default_max_iters(double rho) { return (int64_t)std::nearbyint(std::log(1e-6) / std::log(rho)); }

// the formulas live in cgo_ctl.hpp: one definition for the host engine and the on-device controller
double beta_from_scalars(const cgo_beta_config &b, const Scal &t, double gu_old, double gg_old,
                         double uu_old) {
    const TrialSums ts = trial_sums(t);
    return beta_from_sums(b.kind, b.mu, ts, gu_old, gg_old, uu_old, beta_norms_fast(ts, uu_old));
}

// LinearAlgebra.norm (BLAS.nrm2 / generic_norm2) returns the true 2-norm whenever it is
// representable; sqrt(Σv²) is that value unless the squares over- or underflowed.  The fused
// launches deliver Σv² for free, so only in those regimes a dedicated pass recomputes the norm
// in the scaled form maxabs·sqrt(Σ (v/maxabs)²).
int Solver::robust_norm(double sumsq, int which, double &out) {
    if (sumsq >= 1e-280 && sumsq <= 1e300) { out = std::sqrt(sumsq); return CGO_OK; }
    double m = 0, ss = 0;
    bool has_nan = false;
    if (int rc = be_->scaled_norm_parts(which, last_eval_a_, m, ss, has_nan)) return rc;
    if (has_nan) out = NAN;
    else if (m == 0.0 || std::isinf(m)) out = m;
    else out = m * std::sqrt(ss);
    return CGO_OK;
}

Solver::Solver(VecBackend *be, const cgo_cg_config &cfg, const cgo_ls_config &ls)
    : be_(be), cfg_(cfg), ls_(ls) {}

Solver::Solver(VecBackend *be, const cgo_cg_config &cfg, const cgo_lss_config &lss)
    : be_(be), cfg_(cfg), ls_{}, sys_(true), lss_(lss) {}

// optim.jl:25-47
int Solver::start() {
    Scal s;
    if (cfg_.beta.kind == CGO_BETA_LBFGS) {
        const int P = cfg_.beta.lbfgs_m + 1;  // physical slots
        int rc = be_->lbfgs_alloc(P);
        if (rc) return rc;
        qn_rho_.assign(P, 0.0);
        qn_list_.clear(); qn_free_ = 0; qn_gamma_ = 1.0;
        qn_gram_ = be_->lbfgs_gram_max_pairs() >= cfg_.beta.lbfgs_m;
        qn_SY_.assign((size_t)P * P, 0.0); qn_YY_.assign((size_t)P * P, 0.0);
        qn_sg_.assign(P, 0.0); qn_yg_.assign(P, 0.0);
    }
    if (sys_ && (cfg_.beta.kind >= CGO_BETA_LBFGS || !be_->sys_supported())) return CGO_EINVAL;  // BT <: CGβConfig (:69)
    int rc = be_->init_eval(s);  // f_x = fdf!(df_x, x); info.u = −df_x
    if (rc) return rc;
    if (sys_ && (rc = be_->sys_begin())) return rc;  // x_next = copy(x_initial)  solve_system.jl:82
    total_evals_ = 1;
    f_x_ = s.f;
    f_x0_ = f_x_;                     // optim.jl:31
    gg_ = s.gtgt;
    if ((rc = robust_norm(gg_, 0, norm_df_x_))) return rc;  // optim.jl:26
    dphi0_ = -gg_;                    // g·(−g)
    uu_ = gg_;
    dir_is_neg_grad_ = true;
    a_initial_ = NAN;                 // optim.jl:47
    it_ = 0; iters_ran_ = 0; status_ = CGO_INCOMPLETE;
    ncache_ = 0; finished_ = false; started_ = true;
    qn_trial_done_ = false;
    res_fail_streak_ = 0; res_backoff_ = 0;
    tr_f_.clear(); tr_g_.clear(); tr_a_.clear(); tr_e_.clear(); log_.clear();
    return CGO_OK;
}

void Solver::finish(int64_t iters, int status) {
    iters_ran_ = iters;
    status_ = status;
    finished_ = true;
    ncache_ = 0;
    qn_trial_done_ = false;
    be_->discard_pending();   // (every committed update has run by now: the direction pass that carries a deferred one belongs to the same iteration)
    if (cfg_.trace_enabled) {  // resizetrace!(ret.trace, i)  types.jl:129,148
        tr_f_.resize((size_t)iters); tr_g_.resize((size_t)iters);
        tr_a_.resize((size_t)iters); tr_e_.resize((size_t)iters);
    }
}

double Solver::first_step(double a_initial) const { return ls_first_step(ls_, a_initial); }

void Solver::first_hints(double a0, double (&h)[2]) const { ls_first_hints(ls_, a0, h[0], h[1]); }

// evalϕdϕ!  (cg_utils.jl:4-23): a result the last launch already produced, or one new launch
// that evaluates `a` together with the hinted candidate steps.
int Solver::eval(double a, double &phi, double &dphi, double h1, double h2, double h3, double h4) {
    const double hs[4] = {h1, h2, h3, h4};
    return evaln(a, phi, dphi, hs, 4);
}

int Solver::evaln(double a, double &phi, double &dphi, const double *hs, int nh) {
    int hit = -1;
    for (int j = 0; j < ncache_; ++j)
        if (std::memcmp(&a, &cache_[j].a, sizeof(double)) == 0) { hit = j; break; }
    if (hit >= 0) {
        last_ = cache_[hit].s;
    } else {
        double pts[7] = {a, 0, 0, 0, 0, 0, 0};
        int k = 1;
        // a trial-only launch of a bisection search is usually the last of its line search (measured:
        // config 5 needs a 5th trial in < 3 % of iterations), so its grandchildren would be wasted work:
        // requested step + both candidates only.  solvesystem walks a long geometric sequence: all of them.
        const int mp = sys_ ? be_->max_points() : std::min(be_->max_points(), 3);
        for (int q = 0; q < nh && k < mp; ++q) {
            const double h = hs[q];
            bool ok = std::isfinite(h) && h > 0.0;
            for (int j = 0; ok && j < k; ++j) ok = (h != pts[j]);
            if (ok) pts[k++] = h;
        }
        Scal out[7];
        if (int rc = be_->trial(pts, k, out)) return rc;
        ncache_ = k;
        for (int j = 0; j < k; ++j) cache_[j] = {pts[j], out[j]};
        last_ = out[0];
    }
    last_eval_a_ = a;
    total_evals_++;
    phi = last_.f;
    dphi = last_.gtu;
    if (log_on_) log_.push_back({a, phi, dphi});
    return CGO_OK;
}

// The two bisection line searches live in cgo_ctl.hpp (ls_strong_wolfe_t, ls_wolfe_bisection_t): the same state
// machines run here over Solver::eval (which launches kernels) and on the device over the finished launch's points.
// nocedal.jl:33-209
int Solver::ls_strong_wolfe(double a_initial, LSOut &o) {
    auto ev = [this](double a, double &phi, double &dphi, double h1, double h2, double h3, double h4) {
        return eval(a, phi, dphi, h1, h2, h3, h4);
    };
    return ls_strong_wolfe_t(ls_, f_x_, dphi0_, a_initial, ev, o);
}

// wolfe.jl:171-207 (used by the Backtracking search below; the Wolfe bisection has its own copy in the template)
int Solver::find_feasible(double &a, double lb, int64_t &evals, double &phi, double &dphi,
                          int &flag, double h1, double h2, double h3, double h4) {
    auto ev = [this](double a_, double &phi_, double &dphi_, double g1, double g2, double g3, double g4) {
        return eval(a_, phi_, dphi_, g1, g2, g3, g4);
    };
    return ls_find_feasible_t(ls_, a, lb, evals, phi, dphi, flag, ev, h1, h2, h3, h4);
}

// wolfe.jl:13-165
int Solver::ls_wolfe_bisection(double a_initial, LSOut &o) {
    auto ev = [this](double a, double &phi, double &dphi, double h1, double h2, double h3, double h4) {
        return eval(a, phi, dphi, h1, h2, h3, h4);
    };
    struct Bk {
        Solver *s;
        int is_neg_grad(bool &yes) {
            yes = s->dir_is_neg_grad_;
            if (!yes) {
                double ss = 0;
                if (int rc = s->be_->upg_sumsq(ss)) return rc;
                yes = (std::sqrt(ss) == 0.0);
            }
            return 0;
        }
        int reset_dir(double &uu) {
            Scal sc;
            if (int rc = s->be_->reset_dir(sc)) return rc;
            s->ncache_ = 0;        // u changed: speculative trials along the old direction are void
            uu = sc.uu;            // YuanWeiLuWolfe re-evaluates dot(u,u) on every check (wolfe.jl:240)
            // getβ is called with the RESET info.u afterwards (optim.jl:130-135): its dot(u, g) — SallehAlhawarat's
            // denominator term, cg_flavours.jl:145 — is −g·g, not the dϕ₀ of the direction the search started with.
            // The line search itself keeps its by-value copy of dϕ₀ (not recomputed, wolfe.jl:125-129).
            s->dphi0_ = sc.gu;
            s->dir_is_neg_grad_ = true;
            return 0;
        }
    } bk{this};
    return ls_wolfe_bisection_t(ls_, f_x_, dphi0_, uu_, a_initial, ev, bk, o);
}

// geometric.jl:22-152, restated bug for bug (see oracle/cgo_oracle.c):
//  - the re-evaluation at geometric.jl:77 repeats the trial findfeasiblestepsize! just made at the
//    same step; the launch is skipped (same inputs ⇒ bitwise same sums) but still counted;
//  - :success returns the PREVIOUS (ϕ, a) while the trial state is the last, rejected one.
int Solver::ls_backtracking(double a_initial, LSOut &o) {
    const double phi0 = f_x_, d0 = dphi0_, c1 = ls_.c1, rho = ls_.discount_factor;
    auto armijo = [&](double phi_a, double a) { return armijo_test(c1, phi_a, a, phi0, d0); };  // geometric.jl:164-186
    if (!std::isfinite(phi0)) { o = {phi0, 0.0, 0, CGO_ACCEPTED_NON_FINITE_ITERATE}; return CGO_OK; }
    if (d0 > 0.0) { o = {phi0, 0.0, 0, CGO_NON_DESCENT_SEARCH_DIRECTION}; return CGO_OK; }
    double a = a_initial;
    if (!std::isfinite(a)) a = std::fabs(phi0) / uu_;   // geometric.jl:49-52
    if (!std::isfinite(a)) a = 1.0;                     // geometric.jl:53-56
    int64_t evals = 0;
    double phi = 0, dphi = 0;
    int flag = 0;
    if (int rc = find_feasible(a, 0.0, evals, phi, dphi, flag, a / rho, a * rho, a / rho / rho, a * rho * rho)) return rc;
    if (flag != CGO_FEASIBLE) { o = {phi0, 0.0, 0, CGO_CANNOT_FIND_INITIAL_FEASIBLE_STEP}; return CGO_OK; }
    ++evals;                                            // geometric.jl:77-78 (identical re-evaluation)
    total_evals_++;
    if (log_on_) log_.push_back({a, phi, dphi});
    const bool divide = armijo(phi, a);                 // valid → grow (a/ρ), else shrink (a·ρ)
    double a_prev = a, phi_prev = phi;
    for (int64_t k = 0; k < ls_.max_iters; ++k) {
        a = divide ? a / rho : a * rho;
        if (!std::isfinite(a)) { o = {phi_prev, a_prev, evals, CGO_NON_FINITE_STEP_PROPOSED}; return CGO_OK; }
        if (a == a_prev) { o = {phi_prev, a_prev, evals, CGO_PROPOSED_STEP_SAME_AS_CURRENT_STEP}; return CGO_OK; }
        const double n1 = divide ? a / rho : a * rho, n2 = divide ? n1 / rho : n1 * rho, n3 = divide ? n2 / rho : n2 * rho;
        if (int rc = eval(a, phi, dphi, n1, n2, n3, divide ? n3 / rho : n3 * rho)) return rc;
        ++evals;
        if (!armijo(phi, a)) { o = {phi_prev, a_prev, evals, CGO_SUCCESS}; return CGO_OK; }
        a_prev = a;
        phi_prev = phi;
    }
    o = {phi, a, evals, CGO_LINESEARCH_MAX_ITERS_REACHED};
    return CGO_OK;
}

// the candidate in `slot` becomes the newest stored pair; the slot it evicts (or the next unused
// one) becomes the free slot for the following candidate
void Solver::qn_commit(int slot) {
    const int m = cfg_.beta.lbfgs_m;
    qn_list_.insert(qn_list_.begin(), slot);
    if ((int)qn_list_.size() > m) {
        qn_free_ = qn_list_.back();
        qn_list_.pop_back();
    } else {
        std::vector<char> used(m + 1, 0);
        for (int p : qn_list_) used[p] = 1;
        for (int p = 0; p <= m; ++p) if (!used[p]) { qn_free_ = p; break; }
    }
}

// "updatedir!" of the new QNβConfig: u = −H·g by the two-loop recursion (Nocedal & Wright Alg. 7.4),
// either as 2m chained device launches or — Gram form — on the host from the stored inner products
// followed by one linear-combination launch.
int Solver::qn_direction(Scal &s, double a_trial, Scal *trial) {
    const int c = (int)qn_list_.size(), P = cfg_.beta.lbfgs_m + 1;
    qn_trial_done_ = false;
    if (c == 0) return be_->reset_dir(s);
    const double gamma = qn_gamma_;
    if (!qn_gram_) return be_->lbfgs_direction(qn_list_.data(), qn_rho_.data(), c, gamma, s);
    const int *L = qn_list_.data();
    std::vector<double> a(c, 0.0), cc(c, 0.0), cy(c), cs(c);
    for (int k = 0; k < c; ++k) {           // newest → oldest: α_k = ρ_k · s_k·q,  q = g − Σ_{j<k} α_j y_j
        double sq = qn_sg_[L[k]];
        for (int j = 0; j < k; ++j) sq -= a[j] * qn_SY_[(size_t)L[k] * P + L[j]];
        a[k] = qn_rho_[L[k]] * sq;
    }
    for (int k = c - 1; k >= 0; --k) {      // oldest → newest: β_k = ρ_k · y_k·r,  r = γq + Σ_{j>k} c_j s_j
        double yq = qn_yg_[L[k]];
        for (int j = 0; j < c; ++j) yq -= a[j] * qn_YY_[(size_t)L[k] * P + L[j]];
        double yr = gamma * yq;
        for (int j = c - 1; j > k; --j) yr += cc[j] * qn_SY_[(size_t)L[j] * P + L[k]];
        cc[k] = a[k] - qn_rho_[L[k]] * yr;
    }
    for (int k = 0; k < c; ++k) { cy[k] = gamma * a[k]; cs[k] = -cc[k]; }   // u = −γg + Σ γα_k y_k − Σ c_k s_k
    if (trial && std::isfinite(a_trial) && be_->lbfgs_direction_gram_can_fuse_trial()) {
        qn_trial_done_ = true;
        return be_->lbfgs_direction_gram_trial(L, cy.data(), cs.data(), c, -gamma, a_trial, s, *trial);
    }
    return be_->lbfgs_direction_gram(L, cy.data(), cs.data(), c, -gamma, s);
}

// A slice of whole outer iterations on the device (cgo_resident.hpp): state out, records in.
int Solver::run_resident(int64_t cap, int64_t &done, int &reason) {
    ResConfig c{};
    c.ls = ls_; c.eps = cfg_.eps; c.mu = cfg_.beta.mu; c.beta_kind = cfg_.beta.kind;
    c.npts = be_->max_points() >= 7 ? 7 : (be_->max_points() >= 3 ? 3 : 1);
    c.max_iters = cfg_.max_iters; c.log_on = log_on_ ? 1 : 0;
    ResState s{};
    s.f_x = f_x_; s.gg = gg_; s.norm = norm_df_x_; s.dphi0 = dphi0_; s.uu = uu_; s.a_initial = a_initial_;
    s.it = it_; s.dir_neg = dir_is_neg_grad_ ? 1 : 0;
    s.ncache = ncache_ <= RES_MAXP ? ncache_ : 0;
    for (int j = 0; j < s.ncache; ++j) { s.ca[j] = cache_[j].a; s.cs[j] = trial_sums(cache_[j].s); }
    if (int rc = be_->resident_run(c, s, cap, res_recs_, res_log_)) return rc;
    done = s.done; reason = s.reason;
    if (s.done > 0) {
        f_x_ = s.f_x; gg_ = s.gg; norm_df_x_ = s.norm; dphi0_ = s.dphi0; uu_ = s.uu; a_initial_ = s.a_initial;
        it_ = s.it; dir_is_neg_grad_ = s.dir_neg != 0;
        last_eval_a_ = s.last_a;
        total_evals_ += s.evals;
        if (cfg_.trace_enabled)
            for (int64_t i = 0; i < s.done; ++i) {
                const ResRecord &r = res_recs_[(size_t)i];
                tr_f_.push_back(r.f); tr_g_.push_back(r.norm); tr_a_.push_back(r.a); tr_e_.push_back(r.evals);
            }
        if (log_on_) for (int64_t i = 0; i < s.log_len; ++i) log_.push_back({res_log_[(size_t)i].a, res_log_[(size_t)i].phi, res_log_[(size_t)i].dphi});
        // the trial sums of the pass that accepted the last step wait in the cache, for this engine or the next slice
        ncache_ = s.ncache;
        for (int j = 0; j < s.ncache; ++j) {
            Scal q; const TrialSums &t = s.cs[j];
            q.f = t.f; q.gtu = t.gtu; q.gtgt = t.gtgt; q.gtg = t.gtg; q.yy = t.yy; q.uy = t.uy; q.ygt = t.ygt;
            q.gu = s.dphi0; q.uu = s.uu;
            cache_[j] = {s.ca[j], q};
        }
    }
    return CGO_OK;
}

// optim.jl:50-160
int Solver::iterate(int64_t iters, bool &finished) {
    if (!started_) return CGO_ESTATE;
    if (sys_) return iterate_sys(iters, finished);
    const bool qn = cfg_.beta.kind == CGO_BETA_LBFGS;
    const bool resident = !qn && be_->resident_ready(cfg_, ls_);
    bool host_next = false;
    for (int64_t budget = iters; budget > 0 && !finished_; --budget) {
        const int64_t n = it_ + 1;
        if (n > cfg_.max_iters) { finish(cfg_.max_iters, CGO_MAX_ITERS_REACHED); break; }  // optim.jl:162-169
        if (std::isfinite(f_x_) && std::isfinite(norm_df_x_) && norm_df_x_ < cfg_.eps) {    // optim.jl:53-80
            finish(n - 1, f_x_ <= f_x0_ ? CGO_SUCCESS : CGO_INCREASING_OBJECTIVE);
            break;
        }
        if (resident && !host_next && res_backoff_ == 0 && be_->resident_ready(cfg_, ls_)) {   // (re-asked: a backend can take itself off the resident path)   // as many whole iterations as the slice allows in ONE launch; what it cannot do comes back here
            int64_t done = 0;
            int reason = RES_HOST;
            if (int rc = run_resident(std::min(budget, cfg_.max_iters - it_), done, reason)) return rc;
            if (reason == RES_ERROR) return CGO_ECOMM;
            host_next = (reason == RES_HOST);                         // the iteration after the completed ones needs the host
            if (done > 0) { res_fail_streak_ = 0; budget -= done - 1; continue; }   // (the loop header takes the last one off)
            // done == 0: iteration n needs the host (or the slice could not start): run it below, as always.  A solve whose
            // every iteration needs the host (a rare-path norm each time, say) would pay each line search twice: back off.
            res_fail_streak_ = std::min(res_fail_streak_ + 1, 6);
            res_backoff_ = (int64_t(1) << res_fail_streak_) - 2;      // 0, 2, 6, 14, 30, 62 host-driven iterations before the next attempt
        } else if (res_backoff_ > 0) {
            --res_backoff_;
        }
        host_next = false;
        LSOut o{};
        int rc = (ls_.kind == CGO_LS_STRONG_WOLFE_BISECTION) ? ls_strong_wolfe(a_initial_, o)
                 : (ls_.kind == CGO_LS_WOLFE_BISECTION)      ? ls_wolfe_bisection(a_initial_, o)
                                                             : ls_backtracking(a_initial_, o);
        if (rc) return rc;
        ncache_ = 0;  // x, u are about to change (or the solve ends): cached trials are void
        a_initial_ = o.a;                                              // optim.jl:92
        if (o.status != CGO_SUCCESS) { finish(n - 1, o.status); break; }  // optim.jl:93-104
        // Two-phase objective: g⁺ of the accepted step was not written during the line search.  Under the Gram form of L-BFGS
        // the push can form it itself (one launch and 24 B/element fewer); x and g stay the last good iterate until the
        // non-finite test below has passed (lbfgs_push_commit).
        // … or the push is already paid for: the direction pass speculated on exactly this step and left every inner product.
        VecBackend::GramOut G;
        bool push_first = false;
        if (qn && qn_gram_) {
            if (be_->lbfgs_push_spec(last_eval_a_, o.a, qn_free_, qn_list_.data(), (int)qn_list_.size(), G)) {
                push_first = true;
            } else if (be_->two_phase() && be_->lbfgs_push_materializes(last_eval_a_)) {
                if ((rc = be_->lbfgs_push_gram(last_eval_a_, o.a, qn_free_, qn_list_.data(), (int)qn_list_.size(), G))) return rc;
                if (!G.materialized) return CGO_ESTATE;   // (a backend that said it would must: the plain push has read a g⁺ nobody wrote)
                push_first = true;
            }
        }
        if (push_first) {
            last_.gtgt = G.gtgt;
        } else if (be_->two_phase() || qn_trial_done_) {   // (a trial that rode in a direction pass of an element-wise objective left no g⁺ either)
            if ((rc = be_->materialize(last_))) return rc;
        }
        double norm_df_xp = NAN;                                        // optim.jl:107
        if ((rc = robust_norm(last_.gtgt, 1, norm_df_xp))) return rc;
        if (!std::isfinite(o.phi) || !std::isfinite(norm_df_xp)) {      // optim.jl:108-121
            finish(n - 1, CGO_NON_FINITE_OBJECTIVE_OR_GRADIENT_PROPOSED);
            break;
        }
        double beta = 0.0;                                              // optim.jl:130-135
        if (!qn) {
            const TrialSums ts = trial_sums(last_);
            BetaNorms bn = beta_norms_fast(ts, uu_);
            bn.gt = norm_df_xp;   // the LinearAlgebra.norm of g⁺ computed above (scaled form where Σg⁺² left the safe range)
            if (!beta_norms_fast_ok(cfg_.beta.kind, ts, uu_) && cfg_.beta.kind == CGO_BETA_YUAN_WANG_SHENG) {
                // norm(u), norm(y) of cg_flavours.jl:65 in their scaled form: two rare-path passes each
                if ((rc = robust_norm(uu_, 3, bn.u))) return rc;
                if ((rc = robust_norm(ts.yy, 4, bn.y))) return rc;
            }
            beta = beta_from_sums(cfg_.beta.kind, cfg_.beta.mu, ts, dphi0_, gg_, uu_, bn);
        }
        // optim.jl:136-141 (the three n-vector copies become a pointer swap + in-register xp)
        f_x_ = o.phi;
        norm_df_x_ = norm_df_xp;
        gg_ = last_.gtgt;
        it_ = n;
        if (cfg_.trace_enabled) {                                       // optim.jl:152-159
            tr_f_.push_back(f_x_); tr_g_.push_back(norm_df_x_);
            tr_a_.push_back(o.a); tr_e_.push_back(o.evals);
        }
        const bool will_stop = (n == cfg_.max_iters) ||
            (std::isfinite(f_x_) && std::isfinite(norm_df_x_) && norm_df_x_ < cfg_.eps);
        // x ← info.xp (optim.jl:136): xp is the LAST evaluated trial; equals a* except under Backtracking
        const double a_xp = last_eval_a_;
        Scal s;
        if (qn) {
            const int slot = qn_free_, c = (int)qn_list_.size(), P = cfg_.beta.lbfgs_m + 1;
            double sy = 0, yy = 0;
            if (qn_gram_) {
                if (push_first) rc = be_->lbfgs_push_commit(!will_stop);   // x ← xp, g ← g⁺ (optim.jl:136-139)
                else rc = be_->lbfgs_push_gram(a_xp, o.a, slot, qn_list_.data(), c, G);
                if (rc) return rc;
                sy = G.sy; yy = G.yy;
                for (int j = 0; j < c; ++j) {       // g changed: s_j·g, y_j·g of every stored pair
                    const int pj = qn_list_[j];
                    if (G.y_based) { qn_sg_[pj] += G.sjyn[j]; qn_yg_[pj] += G.yjyn[j]; }   // b·g⁺ = b·g + b·y
                    else { qn_sg_[pj] = G.sjg[j]; qn_yg_[pj] = G.yjg[j]; }
                    if (sy > 0.0) {                 // Gram row/column of the new pair
                        qn_SY_[(size_t)pj * P + slot] = G.sjyn[j];   // s_j·y_new
                        qn_SY_[(size_t)slot * P + pj] = G.yjsn[j];   // s_new·y_j
                        qn_YY_[(size_t)pj * P + slot] = qn_YY_[(size_t)slot * P + pj] = G.yjyn[j];
                    }
                }
                if (sy > 0.0) {
                    qn_SY_[(size_t)slot * P + slot] = sy; qn_YY_[(size_t)slot * P + slot] = yy;
                    qn_sg_[slot] = G.sgn; qn_yg_[slot] = G.ygn;
                }
            } else {
                if ((rc = be_->lbfgs_push(a_xp, o.a, slot, sy, yy))) return rc;
            }
            if (sy > 0.0) {  // curvature pair kept (dropped otherwise; the stored history is untouched)
                qn_rho_[slot] = 1.0 / sy;
                qn_gamma_ = sy / yy;
                qn_commit(slot);
            }
            if (!will_stop) {
                // the first step of the next line search (optim.jl:92 + nocedal.jl:49-52 / wolfe.jl:30-32) is known now: where the
                // backend can, its trial rides in the direction pass (the two-phase log-sum-exp objective: one launch fewer)
                const double a0 = first_step(a_initial_);
                Scal t0;
                if ((rc = qn_direction(s, ls_.kind == CGO_LS_BACKTRACKING ? NAN : a0, &t0))) return rc;
                dphi0_ = s.gu; uu_ = s.uu;
                dir_is_neg_grad_ = qn_list_.empty();
                if (qn_trial_done_) { ncache_ = 1; cache_[0] = {a0, t0}; }
            }
        } else if (will_stop) {
            if ((rc = be_->accept_only(a_xp))) return rc;
        } else if (!std::isfinite(first_step(a_initial_))) {
            if ((rc = be_->accept_dir(a_xp, beta, s))) return rc;       // optim.jl:145
            dphi0_ = s.gu; uu_ = s.uu;
            dir_is_neg_grad_ = (beta == 0.0);
        } else {
            // (Also at the END of an iterate() slice: the trial sums wait in the cache for the next slice.  The fused launch
            //  moves the same bytes as accept + direction alone, so a caller that stops here has lost nothing, and slicing a
            //  solve no longer changes a single launch — a slice boundary used to cost a trial-only launch, ≈ 380 µs at n = 1e8.)
            // the next line search's first step is known now (optim.jl:92 + nocedal.jl:49-52 /
            // wolfe.jl:30-32), and so are the two steps it can ask for second: evaluate all three
            double pts[7] = {0, 0, 0, 0, 0, 0, 0};
            int k;
            if (be_->max_points() >= 5) {
                k = ls_trial_points_n(ls_, first_step(a_initial_), be_->max_points() >= 7 ? 7 : 5, pts);
            } else {
                double p3[3];
                k = ls_trial_points(ls_, first_step(a_initial_), be_->max_points() >= 3, p3);
                for (int j = 0; j < 3; ++j) pts[j] = p3[j];
            }
            Scal out[7];
            if (be_->ctl_depth() > 0 && !be_->two_phase() && ls_.kind != CGO_LS_BACKTRACKING) {
                // line searches that finish inside the launch that started them run on the device without the
                // host (cgo_ctl.hpp); this loop then replays them from the published records
                CtlConfig cc;
                cc.ls = ls_; cc.eps = cfg_.eps; cc.mu = cfg_.beta.mu;
                cc.beta_kind = cfg_.beta.kind; cc.maxp = be_->max_points();
                cc.max_iters = cfg_.max_iters;
                CtlState cs;
                cs.f_x = f_x_; cs.gg = gg_; cs.a_acc = a_xp; cs.beta = beta;
                for (int j = 0; j < CTL_MAXP; ++j) cs.a[j] = pts[j < k ? j : k - 1];
                cs.npts = k; cs.go = 1; cs.it = it_;
                rc = be_->accept_dir_trial_ctl(cc, cs, budget, out);   // (the fused launch of the slice's last iteration included)
            } else {
                rc = be_->accept_dir_trial(a_xp, beta, pts, k, out);
            }
            if (rc) return rc;
            dphi0_ = out[0].gu; uu_ = out[0].uu;
            dir_is_neg_grad_ = (beta == 0.0);
            ncache_ = k;
            for (int j = 0; j < k; ++j) cache_[j] = {pts[j], out[j]};
        }
    }
    finished = finished_;
    return CGO_OK;
}

// solve_system.jl:109-227 — the outer loop of solvesystem, with its line search (:29-56) inlined.
// Per outer iteration: the trials of the line search (three steps s·ρ^i per launch, the first ones
// fused with the previous direction update), ONE projection launch (z, g(z), x_next += m·g(z),
// g⁺ = g(x_next), f, every getβ sum) and ONE direction launch.  The reference spends
// (7k + 3 + getβ + 6) full-vector passes on the same work.
int Solver::iterate_sys(int64_t iters, bool &finished) {
    for (int64_t budget = iters; budget > 0 && !finished_; --budget) {
        const int64_t n = it_ + 1;
        if (n > cfg_.max_iters) { finish(cfg_.max_iters, CGO_MAX_ITERS_REACHED); break; }   // :229-236
        if (norm_df_x_ < cfg_.eps) { finish(n - 1, CGO_SUCCESS); break; }                   // :112-123 (no isfinite test)
        // linesearch! (:29-56); norm_u_sq = dot(u,u) came with the direction launch
        const double s0 = lss_.s, rho = lss_.rho;
        double a = NAN, phi = NAN, dphi = NAN, nrm = NAN;
        int64_t hit = -1;
        int rc;
        for (int64_t i = 0; i < lss_.max_iters; ++i) {
            a = s0 * std::pow(rho, (double)i);                                               // :44
            double nxt[6];
            for (int q = 0; q < 6; ++q) nxt[q] = (i + 1 + q < lss_.max_iters) ? s0 * std::pow(rho, (double)(i + 1 + q)) : NAN;
            if ((rc = evaln(a, phi, dphi, nxt, 6))) return rc;
            if ((rc = robust_norm(last_.gtgt, 1, nrm))) return rc;                           // :49
            if (!(-dphi < lss_.sigma * a * nrm * uu_)) { hit = i; break; }                   // :50-53
        }
        ncache_ = 0;
        if (hit < 0) { finish(n - 1, CGO_LINESEARCH_FAILED); break; }   // :55 throws in the reference; :131-141 is its intent
        if (nrm < cfg_.eps) {                                            // :145-166: the trial point z is the answer
            f_x_ = phi; norm_df_x_ = nrm; it_ = n;
            if (cfg_.trace_enabled) { tr_f_.push_back(phi); tr_g_.push_back(nrm); tr_a_.push_back(a); tr_e_.push_back(hit); }
            if ((rc = be_->accept_only(a))) return rc;                   // results: minimizer = xp, gradient = g(xp)
            finish(n, CGO_SUCCESS);
            break;
        }
        const double m = a * last_.gtu / (nrm * nrm);                    // :248
        Scal p;
        if ((rc = be_->sys_project(a, m, p))) return rc;                 // :169-177
        total_evals_++;
        double nrm_next = NAN;
        if ((rc = robust_norm(p.gtgt, 2, nrm_next))) return rc;
        if (!std::isfinite(p.f) || !std::isfinite(nrm_next)) {           // :178-191
            finish(n - 1, CGO_NON_FINITE_OBJECTIVE_OR_GRADIENT_PROPOSED);
            break;
        }
        if ((rc = be_->sys_commit())) return rc;                         // :194
        f_x_ = p.f;                                                      // :195
        double beta;                                                     // :199-204
        {
            const TrialSums ts = trial_sums(p);
            BetaNorms bn = beta_norms_fast(ts, uu_);
            bn.gt = nrm_next;
            // (norm(u), norm(y) of YuanWangSheng in extreme ranges: the second-iterate layout has no scaled pass for y;
            //  solvesystem keeps the fast form there)
            beta = beta_from_sums(cfg_.beta.kind, cfg_.beta.mu, ts, dphi0_, gg_, uu_, bn);
        }
        gg_ = p.gtgt;
        norm_df_x_ = nrm_next;                                           // :207
        it_ = n;
        if (cfg_.trace_enabled) { tr_f_.push_back(f_x_); tr_g_.push_back(norm_df_x_); tr_a_.push_back(a); tr_e_.push_back(hit); }  // :213-220
        // updatedir! (:210), fused with the first trials of the next line search when one will run
        const bool will_stop = (n == cfg_.max_iters) || (norm_df_x_ < cfg_.eps);
        Scal out[7];
        double pts[7] = {0, 0, 0, 0, 0, 0, 0};
        int k = 0;
        if (!will_stop && budget > 1 && lss_.max_iters > 0) {
            const int kmax = be_->max_points();
            for (int j = 0; j < kmax && j < lss_.max_iters; ++j) {
                const double aj = s0 * std::pow(rho, (double)j);
                if (!(std::isfinite(aj) && aj > 0.0) || (k > 0 && aj == pts[k - 1])) break;
                pts[k++] = aj;
            }
        }
        if (will_stop) continue;  // the loop ends at its next test; u is not part of the results
        if ((rc = be_->dir_trial(beta, pts, k, out))) return rc;
        dphi0_ = out[0].gu; uu_ = out[0].uu;
        ncache_ = k;
        for (int j = 0; j < k; ++j) cache_[j] = {pts[j], out[j]};
    }
    finished = finished_;
    return CGO_OK;
}

}  // namespace cgo
