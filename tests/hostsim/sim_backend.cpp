// sim_backend.cpp — TEST DOUBLE for the device backend.  NOT PRODUCT CODE.
//
// Lets the GPU-less CPU test tier exercise the product's host control plane
// (conjugategradientoptim.jl_amd/csrc/cgo_engine.cpp: outer loop, both line
// searches, β formulas on reduced scalars, trace/result bookkeeping, rank-
// ordered cross-rank sums) against the independent oracle.  It implements the
// VecBackend interface with plain loops on host memory and is linked ONLY into
// tests/hostsim/_build/libcgo_hostsim.so — never into libcgo_hip.so, whose only
// backend is the HIP one and which fails with CGO_ENODEV when no GPU exists.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../../conjugategradientoptim.jl_amd/csrc/cgo_engine.hpp"

using namespace cgo;

namespace {

struct SimComm {
    int rank = 0, world = 1;
    cgo_allgather_fn fn = nullptr;
    void *user = nullptr;
};

// host-closure objective for the following sim_* calls (objective kind 5)
typedef double (*sim_fdf_fn)(void *user, double *g, const double *x, int64_t n);
static sim_fdf_fn g_host_fn = nullptr;
static void *g_host_user = nullptr;

class SimBackend : public VecBackend {
  public:
    SimBackend(int kind, int64_t n_local, int64_t offset, const double *p0, double s0, SimComm c)
        : kind_(kind), n_(n_local), off_(offset), s0_(s0), comm_(c) {
        x_.assign(n_, 0); u_.assign(n_, 0); ga_.assign(n_, 0); gb_.assign(n_, 0);
        if (p0) p0_.assign(p0, p0 + n_);
        g_ = ga_.data(); gt_ = gb_.data();
    }
    int64_t n_local() const override { return n_; }
    int set_x0_host(const double *x0) override { std::memcpy(x_.data(), x0, sizeof(double) * n_); return 0; }
    int set_x0_fill(int, uint64_t, double lo, double) override { for (auto &v : x_) v = lo; return 0; }

    // element-wise objective at xp → f partial and gradient
    void objective(const double *xp, double *g, double &f) const {
        f = 0;
        if (kind_ == 5 /* CGO_OBJ_HOST: the reference's closure contract f = fdf!(g, x) */) {
            f = g_host_fn(g_host_user, g, xp, n_);
        } else if (kind_ == CGO_OBJ_QUAD_DIAG) {
            for (int64_t i = 0; i < n_; ++i) { g[i] = p0_[i] * xp[i]; f += 0.5 * (g[i] * xp[i]); }
        } else if (kind_ == CGO_OBJ_LSE) {   // f = log Σ exp(x_i) + ½λ‖x‖² (single rank in this double); λ = s0_
            double M = -INFINITY, S = 0.0, Q = 0.0;
            for (int64_t i = 0; i < n_; ++i) if (xp[i] > M) M = xp[i];
            for (int64_t i = 0; i < n_; ++i) { S += std::exp(xp[i] - M); Q += xp[i] * xp[i]; }
            f = (M + std::log(S)) + 0.5 * s0_ * Q;
            for (int64_t i = 0; i < n_; ++i) g[i] = std::exp(xp[i] - M) / S + s0_ * xp[i];
        } else if (kind_ == CGO_OBJ_ROSENBROCK_PAIRED) {
            for (int64_t j = 0; j + 1 < n_; j += 2) {
                const double a = xp[j], b = xp[j + 1], t1 = b - a * a, t2 = 1.0 - a;
                f += 100.0 * (t1 * t1) + t2 * t2;
                g[j] = -400.0 * (a * t1) - 2.0 * t2;
                g[j + 1] = 200.0 * t1;
            }
        } else {  // Booth
            const double t1 = xp[0] + 2 * xp[1] - 7, t2 = 2 * xp[0] + xp[1] - 5;
            f = t1 * t1 + t2 * t2;
            g[0] = 2 * t1 + 2 * t2 * 2;
            g[1] = 2 * t1 * 2 + 2 * t2;
        }
    }
    // rank-major all-gather + rank-ordered sum (what fetch_sums does on the device path)
    int reduce(double *v, int count) {
        if (comm_.world == 1) return 0;
        std::vector<double> all((size_t)count * comm_.world);
        if (comm_.fn(comm_.user, v, all.data(), count) != 0) return CGO_ECOMM;
        for (int s = 0; s < count; ++s) {
            double t = 0;
            for (int r = 0; r < comm_.world; ++r) t += all[(size_t)r * count + s];
            v[s] = t;
        }
        return 0;
    }
    void trial_sums(double a, double *s /*7*/) {
        lse_a_ = a;   // (two-phase objective: the last evaluated step)
        std::vector<double> xp(n_);
        for (int64_t i = 0; i < n_; ++i) xp[i] = x_[i] + a * u_[i];
        double f;
        objective(xp.data(), gt_, f);
        s[0] = f; s[1] = s[2] = s[3] = s[4] = s[5] = s[6] = 0;
        for (int64_t i = 0; i < n_; ++i) {
            const double y = gt_[i] - g_[i];
            s[1] += gt_[i] * u_[i]; s[2] += gt_[i] * gt_[i]; s[3] += gt_[i] * g_[i];
            s[4] += y * y; s[5] += u_[i] * y; s[6] += y * gt_[i];
        }
    }
    static void unpack_trial(const double *s, Scal &o) {
        o.f = s[0]; o.gtu = s[1]; o.gtgt = s[2]; o.gtg = s[3]; o.yy = s[4]; o.uy = s[5]; o.ygt = s[6];
    }
    void dir_sums(double beta, bool reset, double *s /*2*/) {
        s[0] = s[1] = 0;
        for (int64_t i = 0; i < n_; ++i) {
            const double un = reset ? -g_[i] : (-g_[i] + beta * u_[i]);
            s[0] += g_[i] * un; s[1] += un * un;
            u_[i] = un;
        }
    }
    int init_eval(Scal &out) override {
        double s[2];
        objective(x_.data(), gt_, s[0]);
        s[1] = 0;
        for (int64_t i = 0; i < n_; ++i) { u_[i] = -gt_[i]; s[1] += gt_[i] * gt_[i]; }
        std::swap(g_, gt_);
        if (int rc = reduce(s, 2)) return rc;
        out = Scal(); out.f = s[0]; out.gtgt = s[1];
        launches_++;
        return 0;
    }
    int max_points() const override { return points_; }
    int trial(const double *a, int k, Scal *out) override {
        if (int rc = pipe_check_idle("trial")) return rc;
        flush_lite();
        spec_valid_ = false; spec_unmat_ = false;   // this launch writes g⁺ of ITS step
        if (two_phase()) {   // like k_lse_stats: ϕ and dϕ only — g⁺ is NOT written (materialize() / the fused push do that for the accepted step)
            std::vector<double> xp(n_), gt(n_);
            for (int64_t i = 0; i < n_; ++i) xp[i] = x_[i] + a[0] * u_[i];
            double f, gtu = 0;
            objective(xp.data(), gt.data(), f);
            for (int64_t i = 0; i < n_; ++i) gtu += gt[i] * u_[i];
            out[0] = Scal(); out[0].f = f; out[0].gtu = gtu;
            lse_a_ = a[0];
            launches_++;
            return 0;
        }
        for (int j = 0; j < k; ++j) {  // one "launch" evaluates all k points
            double s[7];
            trial_sums(a[j], s);
            if (int rc = reduce(s, 7)) return rc;
            unpack_trial(s, out[j]);
        }
        launches_++;
        return 0;
    }
    // x ← x + a·u ; g ← ∇f(x).  Recomputed rather than swapped in: after a multi-point launch the
    // stored trial gradient belongs to the last point evaluated, not necessarily the accepted one
    // (the product's gradient-free kernels recompute g the same way, bit-identically).
    void accept(double a) {
        for (int64_t i = 0; i < n_; ++i) x_[i] = x_[i] + a * u_[i];
        double f;
        objective(x_.data(), g_, f);
    }
    int accept_dir_trial(double a_acc, double beta, const double *a, int k, Scal *out) override {
        double d[2];
        accept(a_acc);
        dir_sums(beta, false, d);
        if (int rc = reduce(d, 2)) return rc;
        for (int j = 0; j < k; ++j) {
            double s[7];
            trial_sums(a[j], s);
            if (int rc = reduce(s, 7)) return rc;
            unpack_trial(s, out[j]);
        }
        out[0].gu = d[0]; out[0].uu = d[1];
        launches_++;
        return 0;
    }
    // Emulation of the on-device controller: the rounds the device would run ahead are executed
    // eagerly here, each armed by the SAME ctl_step() the gfx950 build compiles for the device;
    // the engine then replays them exactly as it does on the GPU.
  public:
    int ctl_depth_ = 0;
    int64_t ctl_rounds_ = 0, ctl_served_ = 0;
  private:
    std::vector<CtlRecord> pipe_;
    size_t pipe_next_ = 0;
  public:
    int ctl_depth() const override { return (comm_.world == 1) ? ctl_depth_ : 0; }
    int pipe_check_idle(const char *what) {
        for (size_t r = pipe_next_; r < pipe_.size(); ++r)
            if (pipe_[r].npts >= 0) { std::fprintf(stderr, "hostsim: %s while controller rounds are pending\n", what); return CGO_ESTATE; }
        pipe_.clear(); pipe_next_ = 0;
        return 0;
    }
    int accept_dir_trial_ctl(const CtlConfig &cc, const CtlState &s0, int64_t rounds, Scal *out) override {
        if (pipe_next_ >= pipe_.size()) {  // nothing published: start a batch from the host's state
            pipe_.clear(); pipe_next_ = 0;
            CtlState st = s0;
            const int64_t R = std::min<int64_t>(rounds, ctl_depth_);
            for (int64_t r = 0; r < R; ++r) {
                CtlRecord rec;
                if (!st.go) { rec = CtlRecord(); rec.npts = -1; pipe_.push_back(rec); continue; }
                Scal o[CTL_MAXP];
                if (int rc = accept_dir_trial(st.a_acc, st.beta, st.a, st.npts, o)) return rc;
                double sums[CTL_NSUMS] = {0};
                const int np = cc.maxp;
                for (int j = 0; j < st.npts; ++j) {
                    double *q = sums + 7 * j;
                    q[0] = o[j].f; q[1] = o[j].gtu; q[2] = o[j].gtgt; q[3] = o[j].gtg; q[4] = o[j].yy; q[5] = o[j].uy; q[6] = o[j].ygt;
                }
                for (int j = st.npts; j < np; ++j) for (int q = 0; q < 7; ++q) sums[7 * j + q] = sums[7 * (st.npts - 1) + q];
                sums[7 * np] = o[0].gu; sums[7 * np + 1] = o[0].uu;
                ctl_step(cc, st, sums, rec);
                pipe_.push_back(rec);
                ctl_rounds_++;
            }
        }
        const CtlRecord &rec = pipe_[pipe_next_++];
        if (rec.npts < 0) {  // the controller had stopped before this round: run it now
            if (int rc = pipe_check_idle("launch")) return rc;
            return accept_dir_trial(s0.a_acc, s0.beta, s0.a, s0.npts, out);
        }
        if (std::memcmp(&rec.a_acc, &s0.a_acc, 8) || std::memcmp(&rec.beta, &s0.beta, 8) || rec.npts != s0.npts ||
            std::memcmp(rec.a, s0.a, 8 * (size_t)s0.npts)) {
            std::fprintf(stderr, "hostsim: controller ran a launch the host did not ask for\n");
            return CGO_ESTATE;
        }
        const int np = cc.maxp;
        for (int j = 0; j < s0.npts; ++j) unpack_trial(rec.sums + 7 * j, out[j]);
        out[0].gu = rec.sums[7 * np]; out[0].uu = rec.sums[7 * np + 1];
        ctl_served_++;
        return 0;
    }
    // ---- resident solver (cgo_resident.hpp): the SAME res_iterate loop the gfx950 build runs in every thread of
    // k_resident, here over this test double's plain loops — so the CPU tier holds the loop to the oracle.
    int resident_on_ = 0;          // 0 off
    int64_t resident_log_cap_ = 1 << 16;
    int64_t res_slices_ = 0, res_iters_ = 0, res_host_ = 0;
    bool resident_ready(const cgo_cg_config &cfg, const cgo_ls_config &ls) const override {
        return resident_on_ && kind_ != 5 && cfg.beta.kind != CGO_BETA_LBFGS &&
               (ls.kind == CGO_LS_STRONG_WOLFE_BISECTION || ls.kind == CGO_LS_WOLFE_BISECTION);
    }
    struct ResVec {
        static constexpr int kNpts = 0;   // trial steps per pass: a run-time value here (ResConfig::npts)
        SimBackend *b;
        bool leader() const { return true; }
        long long clock() const { return 0; }
        static TrialSums ts(const Scal &o) { return TrialSums{o.f, o.gtu, o.gtgt, o.gtg, o.yy, o.uy, o.ygt}; }
        int trial(const double *a, int k, TrialSums *out) {
            Scal o[RES_MAXP];
            if (b->trial(a, k, o)) return 9;
            for (int j = 0; j < k; ++j) out[j] = ts(o[j]);
            return 0;
        }
        int accept_dir_trial(double a_acc, double beta, const double *a, int k, TrialSums *out, double &gu, double &uu) {
            Scal o[RES_MAXP];
            if (k == 0) { if (b->accept_dir(a_acc, beta, o[0])) return 9; }
            else if (b->accept_dir_trial(a_acc, beta, a, k, o)) return 9;
            for (int j = 0; j < k; ++j) out[j] = ts(o[j]);
            gu = o[0].gu; uu = o[0].uu;
            return 0;
        }
    };
    int resident_run(const ResConfig &c, ResState &s, int64_t budget, std::vector<ResRecord> &recs, std::vector<ResLog> &log) override {
        if (int rc = pipe_check_idle("resident slice")) return rc;
        recs.resize((size_t)std::max<int64_t>(budget, 1));
        log.resize((size_t)resident_log_cap_);
        ResVec v{this};
        res_iterate(c, s, v, budget, recs.data(), log.data(), resident_log_cap_);
        res_slices_++; res_iters_ += s.done; if (s.reason == RES_HOST) res_host_++;
        return 0;
    }
    // ---- solvesystem (solve_system.jl) ----
    std::vector<double> xnext_;
    bool sys_supported() const override { return true; }
    int sys_begin() override { xnext_ = x_; return 0; }
    int sys_project(double a, double m, Scal &out) override {
        std::vector<double> z(n_), gz(n_);
        for (int64_t i = 0; i < n_; ++i) z[i] = x_[i] + a * u_[i];
        double fz, s[7];
        objective(z.data(), gz.data(), fz);
        for (int64_t i = 0; i < n_; ++i) xnext_[i] = xnext_[i] + m * gz[i];
        objective(xnext_.data(), gt_, s[0]);
        s[1] = s[2] = s[3] = s[4] = s[5] = s[6] = 0;
        for (int64_t i = 0; i < n_; ++i) {
            const double y = gt_[i] - g_[i];
            s[1] += gt_[i] * u_[i]; s[2] += gt_[i] * gt_[i]; s[3] += gt_[i] * g_[i];
            s[4] += y * y; s[5] += u_[i] * y; s[6] += y * gt_[i];
        }
        if (int rc = reduce(s, 7)) return rc;
        unpack_trial(s, out);
        launches_++;
        return 0;
    }
    int sys_commit() override {
        x_.swap(xnext_);
        double f;
        objective(x_.data(), g_, f);  // g = ∇f(x), as the gradient-free kernels recompute it
        return 0;
    }
    int dir_trial(double beta, const double *a, int k, Scal *out) override {
        double d[2];
        dir_sums(beta, false, d);
        if (int rc = reduce(d, 2)) return rc;
        for (int j = 0; j < k; ++j) {
            double s[7];
            trial_sums(a[j], s);
            if (int rc = reduce(s, 7)) return rc;
            unpack_trial(s, out[j]);
        }
        out[0].gu = d[0]; out[0].uu = d[1];
        launches_++;
        return 0;
    }
    int accept_dir(double a_acc, double beta, Scal &out) override {
        double s[2];
        if (int rc = pipe_check_idle("accept_dir")) return rc;
        accept(a_acc);
        dir_sums(beta, false, s);
        if (int rc = reduce(s, 2)) return rc;
        out.gu = s[0]; out.uu = s[1];
        launches_++;
        return 0;
    }
    int accept_only(double a_acc) override {
        if (int rc = pipe_check_idle("accept_only")) return rc;
        accept(a_acc); launches_++; return 0;
    }
    int reset_dir(Scal &out) override {
        double s[2];
        if (int rc = pipe_check_idle("reset_dir")) return rc;
        flush_lite();
        dir_sums(0, true, s);
        if (int rc = reduce(s, 2)) return rc;
        out.gu = s[0]; out.uu = s[1];
        launches_++;
        return 0;
    }
    int upg_sumsq(double &out) override {
        double s = 0;
        if (int rc = pipe_check_idle("upg_sumsq")) return rc;
        for (int64_t i = 0; i < n_; ++i) { const double t = u_[i] + g_[i]; s += t * t; }
        if (int rc = reduce(&s, 1)) return rc;
        out = s;
        launches_++;
        return 0;
    }
    int lbfgs_alloc(int m) override {
        m_ = m; S_.assign((size_t)m * n_, 0); Y_.assign((size_t)m * n_, 0);
        spec_valid_ = spec_unmat_ = lite_pending_ = lite_deferred_ = fused_pending_ = false;
        return 0;
    }
    int lbfgs_push(double a, double a_s, int slot, double &sy, double &yy) override {
        double *s = &S_[(size_t)slot * n_], *y = &Y_[(size_t)slot * n_];
        double v[2] = {0, 0};
        for (int64_t i = 0; i < n_; ++i) {
            s[i] = a_s * u_[i]; y[i] = gt_[i] - g_[i];
            v[0] += s[i] * y[i]; v[1] += y[i] * y[i];
        }
        for (int64_t i = 0; i < n_; ++i) x_[i] = x_[i] + a * u_[i];
        std::swap(g_, gt_);  // single-point launches only (L-BFGS): gt_ is the adopted trial gradient
        if (int rc = reduce(v, 2)) return rc;
        sy = v[0]; yy = v[1];
        launches_++;
        return 0;
    }
    int lbfgs_gram_max_pairs() const override { return gram_ ? 12 : 0; }
    int lbfgs_push_gram(double a, double a_s, int slot, const int *prev, int count, GramOut &o) override {
        flush_lite();
        const bool fused = lbfgs_push_materializes(a);
        if (fused) {   // g⁺ formed here (objective at x + a·u into gt_); x advances out of place
            fused_x_.resize(n_);
            for (int64_t i = 0; i < n_; ++i) fused_x_[i] = x_[i] + a * u_[i];
            double f;
            objective(fused_x_.data(), gt_, f);
        }
        double *s = &S_[(size_t)slot * n_], *y = &Y_[(size_t)slot * n_];
        std::vector<double> v(4 + 5 * (size_t)count, 0.0);
        for (int64_t i = 0; i < n_; ++i) {
            s[i] = a_s * u_[i]; y[i] = gt_[i] - g_[i];
            v[0] += s[i] * y[i]; v[1] += y[i] * y[i]; v[2] += s[i] * gt_[i]; v[3] += y[i] * gt_[i];
            for (int j = 0; j < count; ++j) {
                const double sj = S_[(size_t)prev[j] * n_ + i], yj = Y_[(size_t)prev[j] * n_ + i];
                v[4 + 5 * j] += sj * gt_[i]; v[5 + 5 * j] += yj * gt_[i]; v[6 + 5 * j] += sj * y[i];
                v[7 + 5 * j] += yj * s[i]; v[8 + 5 * j] += yj * y[i];
            }
        }
        o.materialized = fused;
        if (fused) {
            if (poison_push_ > 0 && ++fused_seen_ == poison_push_) gt_[0] = NAN;   // (test hook: a non-finite g⁺ out of the fused push)
            double gg = 0;
            for (int64_t i = 0; i < n_; ++i) gg += gt_[i] * gt_[i];
            o.gtgt = gg;
            fused_pending_ = true; fused_pushes_++;
        } else {
            for (int64_t i = 0; i < n_; ++i) x_[i] = x_[i] + a * u_[i];
            std::swap(g_, gt_);
        }
        if (int rc = reduce(v.data(), (int)v.size())) return rc;
        o.sy = v[0]; o.yy = v[1]; o.sgn = v[2]; o.ygn = v[3];
        for (int j = 0; j < count; ++j) {
            o.sjg[j] = v[4 + 5 * j]; o.yjg[j] = v[5 + 5 * j]; o.sjyn[j] = v[6 + 5 * j];
            o.yjsn[j] = v[7 + 5 * j]; o.yjyn[j] = v[8 + 5 * j];
        }
        launches_++;
        return 0;
    }
    int lbfgs_direction_gram(const int *slots, const double *cy, const double *cs, int count, double cg,
                             Scal &out) override {
        flush_lite();
        double v[2] = {0, 0};
        for (int64_t i = 0; i < n_; ++i) {
            double r = cg * g_[i];
            for (int j = 0; j < count; ++j) {
                r = r + cy[j] * Y_[(size_t)slots[j] * n_ + i];
                r = r + cs[j] * S_[(size_t)slots[j] * n_ + i];
            }
            u_[i] = r;
            v[0] += g_[i] * r; v[1] += r * r;
        }
        if (int rc = reduce(v, 2)) return rc;
        out.gu = v[0]; out.uu = v[1];
        launches_++;
        return 0;
    }
    // ---- the one-ring-pass protocol of the product's backend (cgo_hip_backend.hip: lbfgs_direction_spec, lbfgs_push_spec,
    // lbfgs_push_commit(direction_follows), flush_lite, materialize), with plain loops: the direction pass also evaluates the
    // first trial of the next line search and takes every inner product of the next push there; g⁺ of that trial is NOT
    // stored; an accepted speculated trial's state update is deferred to the next direction pass (spec_mode_ 2) or applied
    // by lbfgs_push_commit (1); whatever else touches x, g or the ring first applies it (flush_lite).
    bool lbfgs_direction_gram_can_fuse_trial() const override { return spec_mode_ > 0 && gram_ && kind_ != 5 && m_ - 1 <= 10; }
    int lbfgs_direction_gram_trial(const int *slots, const double *cy, const double *cs, int count, double cg, double a_trial,
                                   Scal &dir, Scal &trial) override {
        spec_valid_ = false;
        if (lite_deferred_) { lite_deferred_ = false; apply_lite(); spec_rode_++; }
        std::vector<double> v(8 + 5 * (size_t)count, 0.0), xp(n_), gt(n_);
        for (int64_t i = 0; i < n_; ++i) {
            double r = cg * g_[i];
            for (int j = 0; j < count; ++j) {
                r = r + cy[j] * Y_[(size_t)slots[j] * n_ + i];
                r = r + cs[j] * S_[(size_t)slots[j] * n_ + i];
            }
            u_[i] = r;
            v[5] += g_[i] * r; v[6] += r * r;
            xp[i] = x_[i] + a_trial * r;
        }
        objective(xp.data(), gt.data(), v[0]);
        for (int64_t i = 0; i < n_; ++i) {
            const double y = gt[i] - g_[i];
            v[1] += gt[i] * u_[i]; v[2] += gt[i] * gt[i]; v[3] += y * gt[i]; v[4] += u_[i] * y; v[7] += y * y;
            for (int j = 0; j < count; ++j) {
                const double sj = S_[(size_t)slots[j] * n_ + i], yj = Y_[(size_t)slots[j] * n_ + i];
                v[8 + 5 * j] += sj * gt[i]; v[9 + 5 * j] += yj * gt[i]; v[10 + 5 * j] += sj * y; v[11 + 5 * j] += yj * y; v[12 + 5 * j] += yj * u_[i];
            }
        }
        if (int rc = reduce(v.data(), (int)v.size())) return rc;
        dir.gu = v[5]; dir.uu = v[6];
        trial = Scal(); trial.f = v[0]; trial.gtu = v[1]; trial.gtgt = v[2];
        spec_s_ = v; spec_a_ = a_trial; spec_slots_.assign(slots, slots + count);
        spec_valid_ = true; spec_unmat_ = true;
        lse_a_ = a_trial;   // (two-phase objective: the last evaluated step)
        launches_++;
        return 0;
    }
    bool lbfgs_push_spec(double a_x, double a_s, int slot, const int *prev, int count, GramOut &G) override {
        if (!spec_valid_ || lite_pending_ || lite_deferred_) return false;
        if (std::memcmp(&a_x, &spec_a_, sizeof(double)) != 0 || count != (int)spec_slots_.size()) return false;
        for (int j = 0; j < count; ++j) if (prev[j] != spec_slots_[j]) return false;
        const std::vector<double> &v = spec_s_;
        G.sy = a_s * v[4]; G.yy = v[7]; G.sgn = a_s * v[1]; G.ygn = v[3]; G.gtgt = v[2];
        if (!(G.gtgt >= 1e-280 && G.gtgt <= 1e300) || !std::isfinite(G.sy) || !std::isfinite(G.yy)) return false;
        for (int j = 0; j < count; ++j) {
            G.sjg[j] = v[8 + 5 * j]; G.yjg[j] = v[9 + 5 * j]; G.sjyn[j] = v[10 + 5 * j]; G.yjyn[j] = v[11 + 5 * j]; G.yjsn[j] = a_s * v[12 + 5 * j];
        }
        G.materialized = true; G.y_based = false;
        lite_pending_ = true; lite_a_ = a_x; lite_as_ = a_s; lite_slot_ = slot;
        spec_valid_ = false; spec_unmat_ = false;
        spec_pushes_++;
        return true;
    }
    int lbfgs_push_commit(bool direction_follows) override {
        if (fused_pending_) {   // the fused push: x ← x', g ← g⁺ now
            fused_pending_ = false;
            x_.swap(fused_x_);
            std::swap(g_, gt_);
            return 0;
        }
        if (!lite_pending_) return 0;
        lite_pending_ = false;
        if (direction_follows && spec_mode_ == 2) { lite_deferred_ = true; return 0; }
        apply_lite();
        launches_++;
        return 0;
    }
    bool two_phase() const override { return kind_ == CGO_OBJ_LSE; }
    int materialize(Scal &out) override {
        if (two_phase()) {   // g⁺ of the last evaluated step (k_lse_grad): written now, with the getβ sums
            flush_lite();
            double s7[7];
            trial_sums(lse_a_, s7);
            out.gtgt = s7[2]; out.gtg = s7[3]; out.yy = s7[4]; out.uy = s7[5]; out.ygt = s7[6];
            spec_unmat_ = false;
            launches_++;
            return 0;
        }
        // element-wise objectives: only the trial a direction pass speculated on has no stored g⁺
        if (!spec_unmat_) return 0;
        const double a = spec_a_;
        return trial(&a, 1, &out);
    }
    // the push that forms g⁺ itself (k_lbfgs_push_gram_lse): x and g stay the last good iterate until lbfgs_push_commit
    bool lbfgs_push_materializes(double a_x) override { return fuse_grad_ && two_phase() && gram_ && std::memcmp(&a_x, &lse_a_, 8) == 0; }
    void apply_lite() {   // x ← x + a·u ; g ← ∇f(x) ; s = a_s·u ; y = g⁺ − g  (k_lbfgs_push_lite)
        double *sn = &S_[(size_t)lite_slot_ * n_], *yn = &Y_[(size_t)lite_slot_ * n_];
        for (int64_t i = 0; i < n_; ++i) { x_[i] = x_[i] + lite_a_ * u_[i]; sn[i] = lite_as_ * u_[i]; }
        double f;
        objective(x_.data(), gt_, f);
        for (int64_t i = 0; i < n_; ++i) yn[i] = gt_[i] - g_[i];
        std::swap(g_, gt_);
    }
    void flush_lite() {
        if (!lite_deferred_) return;
        lite_deferred_ = false;
        apply_lite();
        launches_++;
        spec_flushed_++;
    }

    int lbfgs_direction(const int *slots, const double *rho, int count, double gamma, Scal &out) override {
        std::vector<double> r(g_, g_ + n_), alpha(count);
        for (int k = 0; k < count; ++k) {
            const double *s = &S_[(size_t)slots[k] * n_], *y = &Y_[(size_t)slots[k] * n_];
            double d = 0;
            for (int64_t i = 0; i < n_; ++i) d += s[i] * r[i];
            if (int rc = reduce(&d, 1)) return rc;
            alpha[k] = rho[slots[k]] * d;
            for (int64_t i = 0; i < n_; ++i) r[i] = r[i] - alpha[k] * y[i];
        }
        for (int64_t i = 0; i < n_; ++i) r[i] = gamma * r[i];
        for (int k = count - 1; k >= 0; --k) {
            const double *s = &S_[(size_t)slots[k] * n_], *y = &Y_[(size_t)slots[k] * n_];
            double d = 0;
            for (int64_t i = 0; i < n_; ++i) d += y[i] * r[i];
            if (int rc = reduce(&d, 1)) return rc;
            const double c = alpha[k] - rho[slots[k]] * d;
            for (int64_t i = 0; i < n_; ++i) r[i] = r[i] + c * s[i];
        }
        double v[2] = {0, 0};
        for (int64_t i = 0; i < n_; ++i) { u_[i] = -r[i]; v[0] += g_[i] * u_[i]; v[1] += u_[i] * u_[i]; }
        if (int rc = reduce(v, 2)) return rc;
        out.gu = v[0]; out.uu = v[1];
        launches_++;
        return 0;
    }
    int scaled_norm_parts(int which, double a_trial, double &maxabs, double &ss, bool &has_nan) override {
        flush_lite();
        if ((which == 1 || which == 4) && points_ > 1) {  // multi-point launches leave gt_ at the LAST evaluated point: recompute
            double s7[7];
            trial_sums(a_trial, s7);
        }
        std::vector<double> tmp;
        const double *v = which == 3 ? u_.data() : (which ? gt_ : g_);
        if (which == 4) {  // y = g⁺ − g
            tmp.resize(n_);
            for (int64_t i = 0; i < n_; ++i) tmp[i] = gt_[i] - g_[i];
            v = tmp.data();
        }
        double m = 0, nanc = 0;
        for (int64_t i = 0; i < n_; ++i) { const double a = std::fabs(v[i]); if (std::isnan(a)) nanc = 1; if (a > m) m = a; }
        if (comm_.world > 1) {  // max / flag merge over ranks
            std::vector<double> all(2 * (size_t)comm_.world);
            double snd[2] = {m, nanc};
            if (comm_.fn(comm_.user, snd, all.data(), 2) != 0) return CGO_ECOMM;
            for (int r = 0; r < comm_.world; ++r) { if (all[2 * r] > m) m = all[2 * r]; nanc += all[2 * r + 1]; }
        }
        maxabs = m; has_nan = nanc > 0; ss = 0;
        if (has_nan || m == 0.0 || std::isinf(m)) return 0;
        for (int64_t i = 0; i < n_; ++i) { const double r = v[i] / m; ss += r * r; }
        launches_++;
        return reduce(&ss, 1);
    }
    int download(double *x, double *g) override {
        flush_lite();
        if (x) std::memcpy(x, x_.data(), sizeof(double) * n_);
        if (g) std::memcpy(g, g_, sizeof(double) * n_);
        return 0;
    }
    int64_t launches() const override { return launches_; }

  private:
    int kind_;
    int64_t n_, off_;
    double s0_;
    SimComm comm_;
    std::vector<double> x_, u_, ga_, gb_, p0_, S_, Y_;
    double *g_, *gt_;
    int m_ = 0;
    int64_t launches_ = 0;

    bool spec_valid_ = false, spec_unmat_ = false, lite_pending_ = false, lite_deferred_ = false, fused_pending_ = false;
    std::vector<double> fused_x_;
    double lse_a_ = 0;
    std::vector<double> spec_s_;
    std::vector<int> spec_slots_;
    double spec_a_ = 0, lite_a_ = 0, lite_as_ = 0;
    int lite_slot_ = 0;

  public:
    int points_ = 1;
    bool gram_ = true;
    int spec_mode_ = 0;   // 0: two-pass L-BFGS (trial launch + push + direction) · 1: one pass + its own state-update launch · 2: the update rides in the next direction pass
    int64_t spec_pushes_ = 0, spec_rode_ = 0, spec_flushed_ = 0, fused_pushes_ = 0;
    bool fuse_grad_ = true;   // two-phase objective under L-BFGS: the push forms g⁺ itself (off: materialize() + the plain push)
    int poison_push_ = 0, fused_seen_ = 0;
};

static int g_ctl_depth = 0;
static int g_points = 3;
static int64_t g_ctl_rounds = 0, g_ctl_served = 0;
static int g_resident = 0;
static int64_t g_resident_log_cap = 1 << 16;
static int g_lbfgs_spec = 0;
static int g_fuse_grad = 1;
static int g_poison_push = 0;
static int64_t g_fused_pushes = 0;
static int64_t g_spec_pushes = 0, g_spec_rode = 0, g_spec_flushed = 0;
static int64_t g_res_slices = 0, g_res_iters = 0, g_res_host = 0;

}  // namespace

extern "C" {

void sim_set_host_objective(sim_fdf_fn fn, void *user) { g_host_fn = fn; g_host_user = user; }
// depth of the emulated on-device controller for the following sim_minimize calls (0 = off)
void sim_set_ctl_depth(int depth) { g_ctl_depth = depth; }
// trial steps per emulated launch for the following calls: 3 (default) or 5
void sim_set_points(int points) { g_points = points; }
// resident-solver emulation for the following sim_minimize calls (0 = off); log_cap: capacity of a slice's trial log
void sim_set_resident(int on, int64_t log_cap) { g_resident = on; g_resident_log_cap = log_cap > 0 ? log_cap : (1 << 16); }
// slices run / iterations completed inside slices / slices that handed an iteration back to the host
void sim_resident_stats(int64_t *slices, int64_t *iters, int64_t *host) { *slices = g_res_slices; *iters = g_res_iters; *host = g_res_host; }
// one-ring-pass L-BFGS protocol for the following sim_minimize calls: 0 off (two passes + a trial launch), 1 one pass + a
// state-update launch of its own, 2 the state update rides in the next direction pass
void sim_set_lbfgs_spec(int mode) { g_lbfgs_spec = mode; }
// two-phase objective under L-BFGS: 1 (default) the push forms g⁺ itself and x, g change only at lbfgs_push_commit; 0 materialize() + plain push
void sim_set_fuse_grad(int on) { g_fuse_grad = on; }
int64_t sim_fused_pushes(void) { return g_fused_pushes; }
// test hook: the k-th fused push of the following solves yields a NaN in g⁺ (0 = off) — optim.jl:108-121 must then return the last good iterate
void sim_set_poison_push(int k) { g_poison_push = k; }
// pushes paid for by a direction pass / state updates that rode in the next direction pass / that had to be applied early
void sim_lbfgs_spec_stats(int64_t *pushes, int64_t *rode, int64_t *flushed) { *pushes = g_spec_pushes; *rode = g_spec_rode; *flushed = g_spec_flushed; }
// rounds the emulated controller executed / launches the engine was served from its records
void sim_ctl_stats(int64_t *rounds, int64_t *served) { *rounds = g_ctl_rounds; *served = g_ctl_served; }

// Runs the PRODUCT engine (cgo::Solver) over the test-double backend.
// chunk > 0 runs the solve in iterate(chunk) slices to exercise resumability.
int sim_minimize(int obj_kind, int64_t n_local, int64_t offset, const double *p0_local, double s0,
                 const double *x0_local, const cgo_cg_config *cfg, const cgo_ls_config *ls,
                 int rank, int world, cgo_allgather_fn fn, void *user, int64_t chunk,
                 cgo_results *out, int64_t log_cap, double *log_a, double *log_phi, double *log_dphi,
                 int64_t *log_len) {
    std::string why;
    if (int rc = check_cg_config(cfg, why)) return rc;
    if (int rc = check_ls_config(ls, why)) return rc;
    SimComm c; c.rank = rank; c.world = world; c.fn = fn; c.user = user;
    SimBackend be(obj_kind, n_local, offset, p0_local, s0, c);
    be.points_ = (chunk < 0 || cfg->beta.kind == CGO_BETA_LBFGS || obj_kind == CGO_OBJ_LSE) ? 1 : g_points;  // chunk < 0: single-point launches,
    be.gram_ = chunk >= 0;                                                  //            two-loop L-BFGS
    if (chunk < 0) chunk = 0;
    be.ctl_depth_ = g_ctl_depth;
    be.resident_on_ = g_resident; be.resident_log_cap_ = g_resident_log_cap;
    be.spec_mode_ = g_lbfgs_spec;
    be.fuse_grad_ = g_fuse_grad != 0;
    be.poison_push_ = g_poison_push;
    Solver sv(&be, *cfg, *ls);
    sv.set_log_enabled(log_cap > 0);
    be.set_x0_host(x0_local);
    if (int rc = sv.start()) return rc;
    bool fin = false;
    while (!fin)
        if (int rc = sv.iterate(chunk > 0 ? chunk : (int64_t)1 << 40, fin)) return rc;
    out->objective = sv.objective();
    out->iters_ran = sv.iters_ran();
    out->status = sv.status();
    out->total_fdf_evals = sv.total_evals();
    out->total_launches = be.launches();
    g_ctl_rounds = be.ctl_rounds_; g_ctl_served = be.ctl_served_;
    g_res_slices = be.res_slices_; g_res_iters = be.res_iters_; g_res_host = be.res_host_;
    g_spec_pushes = be.spec_pushes_; g_spec_rode = be.spec_rode_; g_spec_flushed = be.spec_flushed_;
    g_fused_pushes = be.fused_pushes_;
    const size_t k = sv.trace_objective().size();
    if (out->trace_objective && k) std::memcpy(out->trace_objective, sv.trace_objective().data(), k * 8);
    if (out->trace_grad_norm && k) std::memcpy(out->trace_grad_norm, sv.trace_grad_norm().data(), k * 8);
    if (out->trace_step_size && k) std::memcpy(out->trace_step_size, sv.trace_step_size().data(), k * 8);
    if (out->trace_objective_evals && k) std::memcpy(out->trace_objective_evals, sv.trace_evals().data(), k * 8);
    be.download(out->minimizer, out->gradient);
    const auto &L = sv.trial_log();
    if (log_len) *log_len = (int64_t)L.size();
    for (int64_t i = 0; i < (int64_t)L.size() && i < log_cap; ++i) {
        log_a[i] = L[i].a; log_phi[i] = L[i].phi; log_dphi[i] = L[i].dphi;
    }
    return 0;
}

// solvesystem over the test double (solve_system.jl:64-253)
int sim_solvesystem(int obj_kind, int64_t n_local, int64_t offset, const double *p0_local, double s0,
                    const double *x0_local, const cgo_cg_config *cfg, const cgo_lss_config *ls,
                    int rank, int world, cgo_allgather_fn fn, void *user, int64_t chunk,
                    cgo_results *out, int64_t log_cap, double *log_a, double *log_phi, double *log_dphi,
                    int64_t *log_len) {
    std::string why;
    if (int rc = check_cg_config(cfg, why)) return rc;
    if (int rc = check_lss_config(ls, why)) return rc;
    SimComm c; c.rank = rank; c.world = world; c.fn = fn; c.user = user;
    SimBackend be(obj_kind, n_local, offset, p0_local, s0, c);
    be.points_ = chunk < 0 ? 1 : g_points;
    if (chunk < 0) chunk = 0;
    Solver sv(&be, *cfg, *ls);
    sv.set_log_enabled(log_cap > 0);
    be.set_x0_host(x0_local);
    if (int rc = sv.start()) return rc;
    bool fin = false;
    while (!fin)
        if (int rc = sv.iterate(chunk > 0 ? chunk : (int64_t)1 << 40, fin)) return rc;
    out->objective = sv.objective();
    out->iters_ran = sv.iters_ran();
    out->status = sv.status();
    out->total_fdf_evals = sv.total_evals();
    out->total_launches = be.launches();
    const size_t k = sv.trace_objective().size();
    if (out->trace_objective && k) std::memcpy(out->trace_objective, sv.trace_objective().data(), k * 8);
    if (out->trace_grad_norm && k) std::memcpy(out->trace_grad_norm, sv.trace_grad_norm().data(), k * 8);
    if (out->trace_step_size && k) std::memcpy(out->trace_step_size, sv.trace_step_size().data(), k * 8);
    if (out->trace_objective_evals && k) std::memcpy(out->trace_objective_evals, sv.trace_evals().data(), k * 8);
    be.download(out->minimizer, out->gradient);
    const auto &L = sv.trial_log();
    if (log_len) *log_len = (int64_t)L.size();
    for (int64_t i = 0; i < (int64_t)L.size() && i < log_cap; ++i) {
        log_a[i] = L[i].a; log_phi[i] = L[i].phi; log_dphi[i] = L[i].dphi;
    }
    return 0;
}

double sim_beta_from_scalars(const cgo_beta_config *b, const double *t7, double gu_old, double gg_old,
                             double uu_old) {
    Scal s;
    s.gtu = t7[0]; s.gtgt = t7[1]; s.gtg = t7[2]; s.yy = t7[3]; s.uy = t7[4]; s.ygt = t7[5];
    return beta_from_scalars(*b, s, gu_old, gg_old, uu_old);
}

const char *sim_status_name(int s) { return status_name(s); }

}  // extern "C"
