// cgo_engine.hpp — host-side iteration engine (device-agnostic control plane).
//
// The engine owns the outer loop of minimizeobjective (reference
// src/engine/optim.jl:6-171) and the two bisection line searches
// (src/linesearch/nocedal.jl:33-209, src/linesearch/wolfe.jl:13-207) as
// scalar state machines.  It never touches an n-vector: all vector work is
// delegated to a VecBackend, whose fused launches hand back a small block of
// globally reduced scalars.  The product's only VecBackend is the HIP one
// (cgo_hip_backend.hip); tests/hostsim/ holds a test double used to exercise
// this control plane on GPU-less machines.
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/cgo.h"
#include "cgo_ctl.hpp"
#include "cgo_resident.hpp"

namespace cgo {

// Reduced scalars of one fused launch (global sums, identical on every rank).
struct Scal {
    // trial part, at xp = x + a·u with gt = ∇f(xp), y = gt − g
    double f = 0;     // ϕ(a)                         cg_utils.jl:19
    double gtu = 0;   // dϕ(a) = gt·u                 cg_utils.jl:20
    double gtgt = 0;  // ‖gt‖²                        optim.jl:107
    double gtg = 0;   // gt·g
    double yy = 0;    // y·y
    double uy = 0;    // u·y
    double ygt = 0;   // y·gt
    // direction part, for the freshly updated u and the current g
    double gu = 0;    // dϕ₀ = g·u                    nocedal.jl:56, wolfe.jl:40
    double uu = 0;    // u·u                          wolfe.jl:240
};

enum KernelKind : int {
    KK_INIT = 0,          // g = ∇f(x), u = −g                       optim.jl:25, cg_flavours.jl:29
    KK_TRIAL,             // evalϕdϕ! + β partial sums                cg_utils.jl:4-23
    KK_ACCEPT_DIR_TRIAL,  // x←xp, g←g⁺, updatedir!, next first trial optim.jl:136-145 + cg_utils.jl
    KK_ACCEPT_DIR,        // x←xp, g←g⁺, updatedir!                   optim.jl:136-145
    KK_ACCEPT_ONLY,       // x←xp, g←g⁺                               optim.jl:136-140
    KK_RESET_DIR,         // u = −g                                   wolfe.jl:129
    KK_UPG_NORM,          // ‖u+g‖²                                   wolfe.jl:123
    KK_LBFGS_PUSH,        // s = a·u, y = g⁺−g, x←xp, g←g⁺            (new QN state update)
    KK_LBFGS_LOOP,        // q ← q + c·v fused with the next dot      (two-loop recursion)
    KK_LBFGS_FINAL,       // u = −r fused with dϕ₀, u·u
    KK_LSE_STATS,         // two-phase objectives, phase 1: ϕ, dϕ of a trial from reductions only
    KK_LSE_GRAD,          // phase 2: materialise g⁺ of the accepted trial + getβ partial sums
    KK_SCALED_NORM,       // max|v| and Σ(v/max)² — LinearAlgebra.norm when Σv² over/underflows
    KK_DIR_TRIAL,         // solvesystem: updatedir! + the first trials of the next line search   solve_system.jl:210,43-46
    KK_SYS_PROJECT,       // solvesystem: x_next += m·g(z); g⁺ = g(x_next); getβ sums              solve_system.jl:169-204
    KK_RESIDENT,          // a slice of WHOLE outer iterations in one launch, state in LDS (cgo_resident.hpp)
    KK_COUNT
};

// Cross-rank exchange of a few doubles.  Every rank ends up with the rank-major
// concatenation, then sums in rank order → bitwise identical scalars everywhere
// → replicated control flow needs no broadcast.
struct Comm {
    int rank = 0, world = 1;
    virtual ~Comm() {}
    virtual int allgather_host(const double *send, double *recv, int count) = 0;
    // device-buffer path (RCCL); returns <0 if unsupported
    virtual int allgather_device(const double *, double *, int, void * /*hipStream_t*/) { return -1; }
    // host shared-memory mailbox (one node): each rank's finalize kernel stores its block + a
    // sequence word straight into its own slot of a segment every rank maps; non-null = available.
    // slot(rank, buf) → host pointer to {double v[64]; uint64 seq; pad}; dev(...) = the device alias.
    virtual int kind() const { return 3; }          // cgo_ctx_comm_info: 1 mailbox, 2 RCCL, 3 host callback
    virtual int ranks_seen() { return world; }
    virtual double *shm_slot_host(int /*rank*/, int /*buf*/) { return nullptr; }
    virtual double *shm_slot_dev(int /*rank*/, int /*buf*/) { return nullptr; }
    // Device mailboxes (one per rank, in that rank's HBM, opened by every peer over hipIpc / xGMI): the finisher of a
    // controller-armed launch stores its block straight into every peer's mailbox and sums the world's blocks itself —
    // no host in the exchange.  connect_devices() is COLLECTIVE (every rank, after all have attached); → 1 usable, 0 not.
    // dev_mailbox(r): rank r's mailbox as THIS device addresses it, [2 buffers][world][72 doubles]; nullptr = not connected.
    virtual int connect_devices() { return 0; }
    virtual double *dev_mailbox(int /*rank*/) { return nullptr; }
};

// Device-vector operations on this rank's shard.  Every method that fills a
// Scal returns GLOBAL sums.  Methods return 0 or a CGO_E* code.
struct VecBackend {
    virtual ~VecBackend() {}
    virtual int64_t n_local() const = 0;
    virtual int set_x0_host(const double *x0) = 0;
    virtual int set_x0_fill(int kind, uint64_t seed, double lo, double hi) = 0;
    // g = ∇f(x); u = −g.  out.f = f(x), out.gtgt = g·g
    virtual int init_eval(Scal &out) = 0;
    // How many trial steps one launch can evaluate (1, or 3 / 5 / 7 for the multi-point CG kernels).
    virtual int max_points() const { return 1; }
    // for each of the k steps a[j]: gt = ∇f(x + a[j]·u) → all trial scalars in out[j]
    virtual int trial(const double *a, int k, Scal *out) = 0;
    // x += a_acc·u; g ← g⁺; u = −g + β·u → gu, uu (in out[0]); then the k trials as above
    virtual int accept_dir_trial(double a_acc, double beta, const double *a, int k, Scal *out) = 0;
    // On-device controller (cgo_ctl.hpp).  ctl_depth() > 0: accept_dir_trial_ctl runs the launch
    // described by `s` like accept_dir_trial, and may run up to `rounds` − 1 FURTHER launches ahead
    // of the host, each armed by ctl_step() on the device from the previous launch's sums; later
    // calls are then answered from the published records after a bit-for-bit check that the
    // device ran exactly the launch the host asks for.
    virtual int ctl_depth() const { return 0; }
    virtual int accept_dir_trial_ctl(const CtlConfig &, const CtlState &s, int64_t /*rounds*/, Scal *out) {
        return accept_dir_trial(s.a_acc, s.beta, s.a, s.npts, out);
    }
    // Resident solver (cgo_resident.hpp): whole outer iterations inside ONE launch — x, u and the parameter vector stay in
    // LDS, every workgroup runs the line search, getβ and the direction update itself (res_iterate) — for the
    // configurations resident_ready() admits (element-wise built-in objective, CG β, a bisection line search, cache-sized
    // shard).  resident_run advances `s` by up to `budget` iterations and returns their records (and trial log).
    virtual bool resident_ready(const cgo_cg_config &, const cgo_ls_config &) const { return false; }
    virtual int resident_run(const ResConfig &, ResState &, int64_t /*budget*/, std::vector<ResRecord> &, std::vector<ResLog> &) { return CGO_EINVAL; }
    // solvesystem (solve_system.jl:64-253).  The second iterate buffer `x_next` (:82) lives in the backend.
    virtual bool sys_supported() const { return false; }
    virtual int sys_begin() { return CGO_EINVAL; }                      // x_next ← x
    // z = x + a·u, g_z = ∇f(z); x_next ← x_next + m·g_z (:239-253); g⁺ = ∇f(x_next) (:177)
    // → out.f = f(x_next) and the getβ sums of g⁺ against g = ∇f(x) and u (:199-204)
    virtual int sys_project(double /*a*/, double /*m*/, Scal &) { return CGO_EINVAL; }
    virtual int sys_commit() { return CGO_EINVAL; }                     // x, x_next = x_next, x (:194)
    // u ← −∇f(x) + β·u → gu, uu (out[0]); then the k trials along the new u (k may be 0)
    virtual int dir_trial(double /*beta*/, const double * /*a*/, int /*k*/, Scal *) { return CGO_EINVAL; }
    // x += a_acc·u; g ⇄ gt; u = −g + β·u → gu, uu
    virtual int accept_dir(double a_acc, double beta, Scal &out) = 0;
    // x += a_acc·u; g ⇄ gt
    virtual int accept_only(double a_acc) = 0;
    // u = −g → gu, uu
    virtual int reset_dir(Scal &out) = 0;
    // Σ (u_i + g_i)²
    virtual int upg_sumsq(double &out) = 0;
    // L-BFGS (new QNβConfig).  The ring has m+1 PHYSICAL slots: the candidate pair of an iteration is
    // written to the one free slot and only becomes part of the history if s·y > 0 (then the oldest
    // slot becomes the free one) — a dropped candidate never clobbers a stored pair.
    // push: s_slot = a_s·u, y_slot = gt − g, x += a_x·u, g ⇄ gt → sy, yy
    // (a_x ≠ a_s only under Backtracking, whose returned step is not the step of xp)
    virtual int lbfgs_push(double a_x, double a_s, int slot, double &sy, double &yy) = 0;
    // two-loop recursion over `count` stored pairs (slots newest→oldest in `slots`);
    // u = −H·g → gu, uu.  rho/gamma are host scalars.
    virtual int lbfgs_direction(const int *slots, const double *rho, int count, double gamma,
                                Scal &out) = 0;
    virtual int lbfgs_alloc(int slots) = 0;
    // "Vector-free" form: all inner products of the two-loop recursion come from ONE pass over
    // the ring (fused with the push) and the direction is ONE linear-combination pass, instead
    // of 2m dependent dot+axpy launches.  0 = not available for this backend/objective.
    virtual int lbfgs_gram_max_pairs() const { return 0; }
    struct GramOut {  // candidate pair (sn, yn), current trial gradient g⁺, stored pairs j
        double sy, yy, sgn, ygn;
        double sjg[16], yjg[16], sjyn[16], yjsn[16], yjyn[16];  // s_j·g⁺, y_j·g⁺, s_j·yn, y_j·sn, y_j·yn
        bool materialized = false;   // the push formed g⁺ itself (two-phase objective): gtgt is valid, lbfgs_push_commit is owed
        bool y_based = false;        // sjg / yjg are not filled: s_j·g⁺ = s_j·g + sjyn[j], y_j·g⁺ = y_j·g + yjyn[j] (lbfgs_push_spec)
        double gtgt = 0.0;
    };
    virtual int lbfgs_push_gram(double, double, int, const int *, int, GramOut &) { return CGO_EINVAL; }
    // Two-phase objective: can the push for the step a_x form g⁺ itself (no materialize() before it)?  Such a push leaves
    // x and g — the last good iterate of optim.jl:108-121 — untouched until lbfgs_push_commit().
    virtual bool lbfgs_push_materializes(double /*a_x*/) { return false; }
    virtual int lbfgs_push_commit(bool /*direction_follows*/) { return CGO_OK; }   // (true: the next call is the direction pass — the state update may ride in it)
    // A solve has ended (or a new one starts): whatever state update was speculated, materialised or deferred but never
    // committed must NOT be applied by a later download — x, g stay the last good iterate (optim.jl:93-121).
    // … or the push may already be paid for: the direction pass that speculated on the step a_x left every inner product
    // (lbfgs_direction_gram_trial of a backend that does so).  true = `out` is filled (y_based), nothing has been launched,
    // lbfgs_push_commit() runs the state update.
    virtual bool lbfgs_push_spec(double /*a_x*/, double /*a_s*/, int /*slot*/, const int * /*prev*/, int /*count*/, GramOut &) { return false; }
    // u = cg·g + Σ_j cy[j]·Y[slots[j]] + cs[j]·S[slots[j]] → gu, uu
    virtual int lbfgs_direction_gram(const int *, const double *, const double *, int, double, Scal &) { return CGO_EINVAL; }
    // … and, where the backend can, the FIRST TRIAL of the next line search in the same pass (its step is known beforehand:
    // optim.jl:92): trial.f, trial.gtu as trial() would return them for {a_trial}.  false = not available: call the plain form.
    virtual bool lbfgs_direction_gram_can_fuse_trial() const { return false; }
    virtual int lbfgs_direction_gram_trial(const int *, const double *, const double *, int, double, double /*a_trial*/, Scal & /*dir*/, Scal & /*trial*/) { return CGO_EINVAL; }
    // Two-phase objectives (not element-wise, e.g. log-sum-exp): trial() returns only ϕ, dϕ;
    // after the line search accepted a step, materialize() writes g⁺ for that step and fills
    // gtgt, gtg, yy, uy, ygt of `out` (f and gtu are left untouched).
    virtual bool two_phase() const { return false; }
    virtual void discard_pending() {}
    virtual int materialize(Scal &) { return 0; }
    virtual int download(double *x, double *g) = 0;
    // rare path of LinearAlgebra.norm: when Σv² over/underflowed, return (max|v_i|, Σ (v_i/max)²,
    // any-NaN) of the current gradient g (which = 0), the trial gradient g⁺ (which = 1), the direction u
    // (which = 3) or y = g⁺ − g (which = 4; both for YuanWangSheng's norm(u)·norm(y), cg_flavours.jl:65)
    // (a_trial: the step of that trial — gradient-free backends must recompute g⁺ from it)
    virtual int scaled_norm_parts(int which, double a_trial, double &maxabs, double &scaled_ss, bool &has_nan) = 0;
    // profiling
    virtual void profile_enable(bool) {}
    virtual void profile_reset() {}
    virtual void profile_get(int, int64_t *launches, double *ms, double *bytes) {
        *launches = 0; *ms = 0; *bytes = 0;
    }
    virtual int64_t launches() const { return 0; }
};

const char *status_name(int s);
const char *kernel_kind_name(int k);
int check_cg_config(const cgo_cg_config *c, std::string &why);
int check_ls_config(const cgo_ls_config *l, std::string &why);
int check_lss_config(const cgo_lss_config *l, std::string &why);
int64_t lss_default_max_iters(double rho);

// getβ(β_config, g_next, g, u) evaluated on the one-pass partial sums.
// gu_old = u·g (the dϕ₀ of the line search just finished), gg_old = g·g.
double beta_from_scalars(const cgo_beta_config &b, const Scal &t, double gu_old, double gg_old,
                         double uu_old);

struct TrialRecord { double a, phi, dphi; };

class Solver {
  public:
    Solver(VecBackend *be, const cgo_cg_config &cfg, const cgo_ls_config &ls);
    Solver(VecBackend *be, const cgo_cg_config &cfg, const cgo_lss_config &lss);  // solvesystem
    int start();                                   // optim.jl:25-47
    int iterate(int64_t iters, bool &finished);    // optim.jl:50-160
    // results (types.jl:107-151)
    double objective() const { return f_x_; }
    int64_t iters_ran() const { return iters_ran_; }
    int status() const { return status_; }
    bool finished() const { return finished_; }
    int64_t total_evals() const { return total_evals_; }
    const std::vector<double> &trace_objective() const { return tr_f_; }
    const std::vector<double> &trace_grad_norm() const { return tr_g_; }
    const std::vector<double> &trace_step_size() const { return tr_a_; }
    const std::vector<int64_t> &trace_evals() const { return tr_e_; }
    const std::vector<TrialRecord> &trial_log() const { return log_; }
    void set_log_enabled(bool on) { log_on_ = on; }
    VecBackend *backend() { return be_; }
    const cgo_cg_config &config() const { return cfg_; }

  private:
    // evalϕdϕ! (cg_utils.jl:4-23).  h1/h2: the (at most two) steps the line search can ask for
    // next, whatever this trial's outcome — evaluated speculatively in the same launch.
    // h3/h4: likelier grandchildren, used by 5-point launches; evaln: any number of hints (solvesystem).
    int eval(double a, double &phi, double &dphi, double h1 = NAN, double h2 = NAN, double h3 = NAN, double h4 = NAN);
    int evaln(double a, double &phi, double &dphi, const double *hints, int nh);
    void first_hints(double a0, double (&h)[2]) const;
    int ls_strong_wolfe(double a_initial, LSOut &o);       // nocedal.jl:33-158
    int ls_wolfe_bisection(double a_initial, LSOut &o);    // wolfe.jl:13-165
    int ls_backtracking(double a_initial, LSOut &o);       // geometric.jl:22-152
    int find_feasible(double &a, double lb, int64_t &evals, double &phi, double &dphi,
                      int &flag, double h1 = NAN, double h2 = NAN, double h3 = NAN, double h4 = NAN);  // wolfe.jl:171-207
    double first_step(double a_initial) const;             // nocedal.jl:49-52 / wolfe.jl:30-32
    int robust_norm(double sumsq, int which, double &out); // LinearAlgebra.norm semantics
    void finish(int64_t iters, int status);
    int iterate_sys(int64_t iters, bool &finished);        // solve_system.jl:109-227
    int run_resident(int64_t cap, int64_t &done, int &reason);   // a slice of whole iterations on the device (cgo_resident.hpp)
    std::vector<ResRecord> res_recs_;
    int res_fail_streak_ = 0;      // slices in a row that completed nothing
    int64_t res_backoff_ = 0;      // host-driven iterations left before the next slice is tried
    std::vector<ResLog> res_log_;

    VecBackend *be_;
    cgo_cg_config cfg_;
    cgo_ls_config ls_;
    bool sys_ = false;       // solvesystem instead of minimizeobjective
    cgo_lss_config lss_{};
    bool started_ = false, finished_ = false;
    // optim.jl loop state
    double f_x_ = NAN, f_x0_ = NAN, norm_df_x_ = NAN, gg_ = NAN;
    double a_initial_ = NAN;
    int64_t it_ = 0;  // completed outer iterations
    int64_t iters_ran_ = 0;
    int status_ = CGO_INCOMPLETE;
    // line-search inputs produced by the last direction launch
    double dphi0_ = NAN, uu_ = NAN;
    bool dir_is_neg_grad_ = true;  // u ≡ −g known by construction (wolfe.jl:123 shortcut)
    // trial results already on the host: the speculative points of the last launch
    struct Cached { double a; Scal s; };
    Cached cache_[7];
    int ncache_ = 0;
    Scal last_;  // scalars of the most recent trial
    double last_eval_a_ = NAN;  // its step: the xp the reference's info.xp/df_xp hold (≠ a* under Backtracking)
    int64_t total_evals_ = 0;
    // L-BFGS host state (physical slots 0..m)
    std::vector<int> qn_list_;   // stored pairs, newest → oldest
    int qn_free_ = 0;            // slot the next candidate pair is written to
    std::vector<double> qn_rho_; // by slot
    double qn_gamma_ = 1.0;
    bool qn_gram_ = false;
    std::vector<double> qn_SY_, qn_YY_, qn_sg_, qn_yg_;  // Gram blocks by physical slot, stride m+1
    int qn_direction(Scal &s, double a_trial = NAN, Scal *trial = nullptr);   // trial != nullptr: fuse the first trial at a_trial where the backend can (*trial_done)
    bool qn_trial_done_ = false;
    void qn_commit(int slot);
    // trace (types.jl:17-23)
    std::vector<double> tr_f_, tr_g_, tr_a_;
    std::vector<int64_t> tr_e_;
    std::vector<TrialRecord> log_;
    bool log_on_ = false;
};

}  // namespace cgo
