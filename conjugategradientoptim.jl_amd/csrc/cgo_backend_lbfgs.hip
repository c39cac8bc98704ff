// cgo_backend_lbfgs.hip — the two-phase log-sum-exp objective and the L-BFGS ring in HBM: Gram form, one ring pass per iteration,
// chained two-loop.
#include "cgo_backend_internal.hpp"

#include "cgo_kernels.hip.hpp"
#include "cgo_kernels_lse.hip.hpp"
#include "cgo_kernels_cg.hip.hpp"

namespace cgo {

using namespace dev;

// ---- two-phase objective (log-sum-exp) --------------------------------------------------
template <int MODE>
static int launch_lse_stats(const LseParams &P, bool big, bool ref, int grid, hipStream_t st) {
    if (ref) {
        if (big) k_lse_stats<MODE, true, true><<<grid, BLOCK, 0, st>>>(P);
        else k_lse_stats<MODE, false, true><<<grid, BLOCK, 0, st>>>(P);
    } else {
        if (big) k_lse_stats<MODE, true, false><<<grid, BLOCK, 0, st>>>(P);
        else k_lse_stats<MODE, false, false><<<grid, BLOCK, 0, st>>>(P);
    }
    return 0;
}

int HipBackend::lse_stats(int mode, double a_acc, double beta, double a_trial, Scal &out, bool dir) {
    if (int rc = flush_lite()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    const int64_t n = obj_->n_local;
    LseParams P;
    P.x = xc_; P.u = u_.p; P.g = g_; P.gt = gt_; P.n = n;
    P.a_acc = a_acc; P.beta = beta; P.a_trial = a_trial; P.lambda = obj_->s0; P.M = 0; P.S = 1;
    P.partials = ctx_->partials;
    // Fixed-reference form (k_lse_stats<…, REF>): the reference is lse of the last point evaluated on this line (for the fused
    // accept + direction + trial launch: of the iterate being accepted).  Not for the very first evaluation (no reference yet).
    const bool ref = pol_.lse_fixed_reference != 0 && mode != LM_NOU && lse_have_;
    const double Mr = ref ? lse_M_ + std::log(lse_S_) : 0.0;
    if (ref) P.M = Mr;
    const double nvec = (mode == LM_NOU) ? 1.0 : (mode == 0 ? 2.0 : 5.0);
    const double bytes = 8.0 * (double)n * nvec;
    const bool big = bytes > big_bytes(mode == 0 || mode == LM_NOU);
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LSE_STATS)) return rc;
    if (mode == 0) launch_lse_stats<0>(P, big, ref, grid, st);
    else if (mode == LM_NOU) launch_lse_stats<LM_NOU>(P, big, ref, grid, st);
    else launch_lse_stats<LM_ACCEPT | LM_DIR>(P, big, ref, grid, st);
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_launch(ctx_, grid, !ref)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s, ref ? MERGE_SUM : MERGE_LSE)) return rc;
    if (prof_on_) prof_commit(KK_LSE_STATS, bytes);
    if (dir) { out.gu = s[S_GU]; out.uu = s[S_UU]; }
    if (ref) {
        const double Sp = s[L_S];
        if (!(Sp >= 1e-280 && Sp <= 1e280) || !std::isfinite(s[L_T])) {   // the trial is far from the reference: take it again from its own maximum
            lse_have_ = false;                                               // (x, u are already updated if this was a fused launch: a plain trial now)
            Scal t;
            if (int rc = lse_stats(0, 0, 0, a_trial, t, false)) return rc;
            out.f = t.f; out.gtu = t.gtu;
            return CGO_OK;
        }
        lse_a_ = a_trial; lse_M_ = Mr; lse_S_ = Sp;   // (M_r, S') describe xp as well as its own (max, Σ) would
        out.f = (Mr + std::log(Sp)) + 0.5 * obj_->s0 * s[L_Q];
        out.gtu = s[L_T] / Sp + obj_->s0 * s[L_R];
        return CGO_OK;
    }
    lse_a_ = a_trial; lse_M_ = s[L_M]; lse_S_ = s[L_S];
    lse_have_ = std::isfinite(lse_M_) && lse_S_ > 0.0 && std::isfinite(lse_S_);
    out.f = (s[L_M] + std::log(s[L_S])) + 0.5 * obj_->s0 * s[L_Q];  // ϕ = lse + ½λ‖xp‖²
    out.gtu = s[L_T] / s[L_S] + obj_->s0 * s[L_R];                   // dϕ = softmax·u + λ xp·u
    return CGO_OK;
}

int HipBackend::lse_grad(bool init, double a, Scal &out) {
    if (int rc = flush_lite()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    const int64_t n = obj_->n_local;
    LseParams P;
    P.x = xc_; P.u = u_.p; P.g = g_; P.gt = gt_; P.n = n;
    P.a_acc = 0; P.beta = 0; P.a_trial = a; P.lambda = obj_->s0; P.M = lse_M_; P.S = lse_S_;
    P.partials = ctx_->partials;
    const bool beta = need_beta_ && !init;
    const double bytes = 8.0 * (double)n * (init ? 3.0 : (beta ? 4.0 : 3.0));
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LSE_GRAD)) return rc;
    if (init) { if (big) k_lse_grad<false, true, true><<<grid, BLOCK, 0, st>>>(P); else k_lse_grad<false, true, false><<<grid, BLOCK, 0, st>>>(P); }
    else if (beta) { if (big) k_lse_grad<true, false, true><<<grid, BLOCK, 0, st>>>(P); else k_lse_grad<true, false, false><<<grid, BLOCK, 0, st>>>(P); }
    else { if (big) k_lse_grad<false, false, true><<<grid, BLOCK, 0, st>>>(P); else k_lse_grad<false, false, false><<<grid, BLOCK, 0, st>>>(P); }
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_launch(ctx_, grid, false)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s)) return rc;
    if (prof_on_) prof_commit(KK_LSE_GRAD, bytes);
    out.gtgt = s[S_GTGT]; out.gtg = s[S_GTG]; out.yy = s[S_YY]; out.uy = s[S_UY]; out.ygt = s[S_YGT];
    return CGO_OK;
}

int HipBackend::materialize(Scal &out) {
    if (obj_->two_phase()) return lse_grad(false, lse_a_, out);
    if (!spec_unmat_) return CGO_OK;   // element-wise objectives: every trial launch writes its g⁺ …
    const double a = spec_a_;          // … except the trial a direction pass speculated on, when its sums could not be used for the push
    return trial(&a, 1, &out);
}

// ---- L-BFGS ring in HBM ------------------------------------------------------------------
int HipBackend::lbfgs_alloc(int m) {  // m = number of physical slots (history + 1)
    HIPCHK(hipSetDevice(ctx_->device));
    qn_m_ = m;
    gram_on_ = pol_.lbfgs_form != 4;     // 4: chained two-loop launches
    if (m - 1 > GRAM_MAXC) gram_on_ = false;
    const size_t n = (size_t)obj_->n_local;
    if (int rc = qn_S_.alloc(ring_ld((long long)n) * (size_t)m)) return rc;   // slots start on 128-B lines (ring_ld)
    if (int rc = qn_Y_.alloc(ring_ld((long long)n) * (size_t)m)) return rc;
    if (qn_alpha_dev_) (void)hipFree(qn_alpha_dev_);
    HIPCHK(hipMalloc((void **)&qn_alpha_dev_, sizeof(double) * 64));
    qn_sgt_slot_ = -1;
    push_pending_ = false; push_lite_pending_ = false; spec_valid_ = false;
    {   // lbfgs_form 0 / 1: one ring pass, state update riding in the next pass · 2: its own launch · 3: Gram form, two passes
      const bool capable = obj_->two_phase() || (!rmode_ && (obj_->kind == CGO_OBJ_QUAD_DIAG || obj_->kind == CGO_OBJ_ROSENBROCK_PAIRED ||
                                                             (obj_->kind == CGO_OBJ_USER && obj_->rtc && obj_->rtc->spec(false, false))));
      spec_on_ = pol_.lbfgs_form <= 2 && gram_on_ && capable; spec_fuse_push_ = pol_.lbfgs_form != 2; }
    spec_unmat_ = false;
    lite_deferred_ = false;
    // the second iterate buffer of the fused push (lbfgs_push_materializes); a rank of a sharded solve that cannot have it
    // fails here rather than falling out of step with its peers, a single rank just keeps the two-launch form
    fuse_grad_ = pol_.lbfgs_fuse_grad != 0;
    if (fuse_grad_ && gram_on_ && obj_->two_phase() && m - 1 <= GRAM_MAXC_LSE && !x2_.p) {
        const int rc = x2_.alloc(n);
        if (rc != CGO_OK && ctx_->world() > 1) return rc;
        if (rc != CGO_OK) (void)hipGetLastError();
    }
    return CGO_OK;
}

// ---- Gram ("vector-free") form of the L-BFGS update ------------------------------------------
// Log-sum-exp objective: the push forms g⁺ of the accepted trial itself (k_lbfgs_push_gram<…, true>) — no k_lse_grad launch.
// Needs a second iterate buffer (x advances out of place until lbfgs_push_commit) and a free row slot for Σ g⁺² (m ≤ 11).
// CGO_LBFGS_FUSE_GRAD=0 keeps materialize() + the plain push (A/B).
bool HipBackend::lbfgs_push_materializes(double a_x) {
    if (!fuse_grad_ || !gram_on_ || !obj_->two_phase() || qn_m_ - 1 > GRAM_MAXC_LSE) return false;
    if (std::memcmp(&a_x, &lse_a_, sizeof(double)) != 0) return false;   // the statistics at hand are those of another step
    return x2_.p != nullptr;   // (lbfgs_alloc: every rank has it or the solve did not start — the ranks' launch sequences must agree)
}

// direction_follows: the caller's next call is the direction of the following iteration — a speculated push then rides in that
// pass (k_lbfgs_combine_spec<…, PUSH>) instead of a launch of its own.  Whatever else touches x, g or the ring first
// (flush_lite at the head of every such entry point) runs the state update as its own launch.
int HipBackend::lbfgs_push_commit(bool direction_follows) {
    if (push_lite_pending_) {
        push_lite_pending_ = false;
        if (direction_follows && spec_fuse_push_) { lite_deferred_ = true; return CGO_OK; }
        return lbfgs_push_lite();
    }
    if (!push_pending_) return CGO_OK;
    push_pending_ = false;
    xc_ = push_xo_;
    std::swap(g_, gt_);  // g ← g⁺
    return CGO_OK;
}

int HipBackend::lbfgs_push_gram(double a_x, double a_s, int slot, const int *prev, int count, GramOut &out) {
    if (int rc = flush_lite()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    if (count > GRAM_MAXC) { set_error("internal: Gram form limited to 12 pairs"); return CGO_EINVAL; }
    const bool fused = lbfgs_push_materializes(a_x) && count <= GRAM_MAXC_LSE;
    const int64_t n = obj_->n_local;
    GramPushParams P;
    P.x = xc_; P.u = u_.p; P.g = g_; P.gt = gt_; P.S = qn_S_.p; P.Y = qn_Y_.p;
    P.n = n; P.a = a_x; P.a_s = a_s; P.slot = slot; P.count = count; P.partials = ctx_->partials;
    for (int j = 0; j < GRAM_MAXC; ++j) P.prev[j] = j < count ? prev[j] : 0;
    GramLseParams L{};
    if (fused) {
        push_xo_ = (xc_ == x_.p) ? x2_.p : x_.p;
        L.xo = push_xo_; L.gt_out = gt_; L.M = lse_M_; L.S = lse_S_; L.lambda = obj_->s0;
    }
    const double bytes = 8.0 * (double)n * (7.0 + 2.0 * count);   // fused: R x,u,g + ring, W x',s,y,g⁺ — the same count, g⁺ written instead of read
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LBFGS_PUSH)) return rc;
    if (fused) {
        if (big) k_lbfgs_push_gram_lse<true><<<grid, BLOCK, 0, st>>>(P, L);
        else k_lbfgs_push_gram_lse<false><<<grid, BLOCK, 0, st>>>(P, L);
    } else {
        if (big) k_lbfgs_push_gram<true><<<grid, BLOCK, 0, st>>>(P);
        else k_lbfgs_push_gram<false><<<grid, BLOCK, 0, st>>>(P);
    }
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_rows(ctx_, grid, NG)) return rc;
    double s[NG];
    if (int rc = fetch_sums(ctx_, s, MERGE_SUM, NG)) return rc;
    if (prof_on_) prof_commit(KK_LBFGS_PUSH, bytes);
    out.sy = s[0]; out.yy = s[1]; out.sgn = s[2]; out.ygn = s[3];
    for (int j = 0; j < count; ++j) {
        out.sjg[j] = s[4 + 5 * j]; out.yjg[j] = s[5 + 5 * j]; out.sjyn[j] = s[6 + 5 * j];
        out.yjsn[j] = s[7 + 5 * j]; out.yjyn[j] = s[8 + 5 * j];
    }
    qn_sgt_slot_ = -1;
    out.materialized = fused;
    push_counts_[fused ? 1 : 2]++;
    if (fused) {          // x, g stay the last good iterate until the caller has seen ‖g⁺‖ (optim.jl:107-121): lbfgs_push_commit
        out.gtgt = s[GRAM_GTGT];
        push_pending_ = true;
    } else {
        std::swap(g_, gt_);  // g ← g⁺
    }
    return CGO_OK;
}

int HipBackend::lbfgs_direction_gram(const int *slots, const double *cy, const double *cs, int count, double cg,
                                     Scal &out) {
    if (int rc = flush_lite()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    if (count > GRAM_MAXC) { set_error("internal: Gram form limited to 12 pairs"); return CGO_EINVAL; }
    const int64_t n = obj_->n_local;
    GramDirParams P;
    P.g = g_; P.u = u_.p; P.S = qn_S_.p; P.Y = qn_Y_.p; P.n = n; P.count = count; P.cg = cg;
    P.partials = ctx_->partials;
    for (int j = 0; j < GRAM_MAXC; ++j) {
        P.slots[j] = j < count ? slots[j] : 0;
        P.cy[j] = j < count ? cy[j] : 0.0;
        P.cs[j] = j < count ? cs[j] : 0.0;
    }
    const double bytes = 8.0 * (double)n * (2.0 + 2.0 * count);
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LBFGS_FINAL)) return rc;
    if (big) k_lbfgs_combine<true><<<grid, BLOCK, 0, st>>>(P);
    else k_lbfgs_combine<false><<<grid, BLOCK, 0, st>>>(P);
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_rows(ctx_, grid, NS)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s)) return rc;
    if (prof_on_) prof_commit(KK_LBFGS_FINAL, bytes);
    out.gu = s[S_GU]; out.uu = s[S_UU];
    return CGO_OK;
}

// The direction pass of the Gram form fused with phase 1 of the next line search's first trial (log-sum-exp objective:
// k_lbfgs_combine_lse).  CGO_LBFGS_FUSE_TRIAL=0 keeps the two launches (A/B).
bool HipBackend::lbfgs_direction_gram_can_fuse_trial() const {
    return pol_.lbfgs_fuse_trial != 0 && gram_on_ && (obj_->two_phase() || (spec_on_ && qn_m_ - 1 <= SPEC_MAXC));
}

int HipBackend::lbfgs_direction_gram_trial(const int *slots, const double *cy, const double *cs, int count, double cg, double a_trial,
                                           Scal &dir, Scal &trial) {
    HIPCHK(hipSetDevice(ctx_->device));
    if (count > GRAM_MAXC) { set_error("internal: Gram form limited to 12 pairs"); return CGO_EINVAL; }
    spec_valid_ = false;
    if (spec_on_ && count <= SPEC_MAXC && qn_m_ - 1 <= SPEC_MAXC) return lbfgs_direction_spec(slots, cy, cs, count, cg, a_trial, dir, trial);
    if (int rc = flush_lite()) return rc;
    if (!obj_->two_phase()) { set_error("internal: k_lbfgs_combine_lse is the log-sum-exp objective's"); return CGO_EINVAL; }
    const int64_t n = obj_->n_local;
    GramDirParams P;
    P.g = g_; P.u = u_.p; P.S = qn_S_.p; P.Y = qn_Y_.p; P.n = n; P.count = count; P.cg = cg;
    P.partials = ctx_->partials;
    for (int j = 0; j < GRAM_MAXC; ++j) {
        P.slots[j] = j < count ? slots[j] : 0;
        P.cy[j] = j < count ? cy[j] : 0.0;
        P.cs[j] = j < count ? cs[j] : 0.0;
    }
    const double bytes = 8.0 * (double)n * (3.0 + 2.0 * count);   // g, x, the ring / u
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LBFGS_FINAL)) return rc;
    if (big) k_lbfgs_combine_lse<true><<<grid, BLOCK, 0, st>>>(P, xc_, a_trial);
    else k_lbfgs_combine_lse<false><<<grid, BLOCK, 0, st>>>(P, xc_, a_trial);
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_launch(ctx_, grid, true)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s, MERGE_LSE)) return rc;
    if (prof_on_) prof_commit(KK_LBFGS_FINAL, bytes);
    dir.gu = s[S_GU]; dir.uu = s[S_UU];
    lse_a_ = a_trial; lse_M_ = s[L_M]; lse_S_ = s[L_S];
    lse_have_ = std::isfinite(lse_M_) && lse_S_ > 0.0 && std::isfinite(lse_S_);
    trial = Scal();
    trial.f = (s[L_M] + std::log(s[L_S])) + 0.5 * obj_->s0 * s[L_Q];   // as lse_stats: ϕ = lse + ½λ‖xp‖², dϕ = softmax·u + λ xp·u
    trial.gtu = s[L_T] / s[L_S] + obj_->s0 * s[L_R];
    return CGO_OK;
}

// ---- one ring pass per outer iteration: direction + first trial + every inner product of the NEXT push, taken at that trial
// (k_lbfgs_combine_spec; CGO_LBFGS_SPEC=0 keeps the two-pass form) ---------------------------------------------------------
template <class Obj>
static void launch_spec(bool big, bool push, int grid, hipStream_t st, const GramDirParams &P, const double *x, double a_trial, const SpecParams &Q, const SpecPush &U) {
    if (push) {
        if (big) k_lbfgs_combine_spec<Obj, true, true><<<grid, BLOCK, 0, st>>>(P, x, a_trial, Q, U);
        else k_lbfgs_combine_spec<Obj, false, true><<<grid, BLOCK, 0, st>>>(P, x, a_trial, Q, U);
    } else {
        if (big) k_lbfgs_combine_spec<Obj, true, false><<<grid, BLOCK, 0, st>>>(P, x, a_trial, Q, U);
        else k_lbfgs_combine_spec<Obj, false, false><<<grid, BLOCK, 0, st>>>(P, x, a_trial, Q, U);
    }
}
template <class Obj>
static void launch_lite(bool big, int grid, hipStream_t st, double *x, const double *u, double *g, double *sn, double *yn, const double *p0, long long n,
                        double a, double a_s, double M, double S, double lambda) {
    if (big) k_lbfgs_push_lite<Obj, true><<<grid, BLOCK, 0, st>>>(x, u, g, sn, yn, p0, n, a, a_s, M, S, lambda);
    else k_lbfgs_push_lite<Obj, false><<<grid, BLOCK, 0, st>>>(x, u, g, sn, yn, p0, n, a, a_s, M, S, lambda);
}

int HipBackend::lbfgs_direction_spec(const int *slots, const double *cy, const double *cs, int count, double cg, double a_trial,
                                     Scal &dir, Scal &trial) {
    const int64_t n = obj_->n_local;
    GramDirParams P;
    P.g = g_; P.u = u_.p; P.S = qn_S_.p; P.Y = qn_Y_.p; P.n = n; P.count = count; P.cg = cg;
    P.partials = ctx_->partials;
    for (int j = 0; j < GRAM_MAXC; ++j) {
        P.slots[j] = j < count ? slots[j] : 0;
        P.cy[j] = j < count ? cy[j] : 0.0;
        P.cs[j] = j < count ? cs[j] : 0.0;
    }
    // The reference: lse(x) itself, from the statistics of the current iterate (the last evaluated trial was accepted as x) —
    // then e_i = exp(xp_i − M_r) ≤ 1 at the maximum of x, S_r = Σ exp(x_i − M_r) = 1 up to rounding (κ = S_r/S' takes care of
    // the rest: ANY reference gives the same g⁺ = κ·p + λ·xp), and S' = exp(lse(xp) − lse(x)) is the change of the log-sum-exp
    // along the step.  The reference follows the iterate, whichever kernel produced its statistics.
    const bool lse = obj_->two_phase();
    const double Mr = lse ? lse_M_ + std::log(lse_S_) : 0.0, Sr = 1.0;
    SpecParams Q{Mr, 1.0 / Sr, obj_->s0, obj_->p0.p};
    const double hp = obj_->uses_param() ? 1.0 : 0.0;
    // the state update of the accepted speculated trial, if it was left to this pass (lbfgs_push_commit(direction_follows))
    const bool push = lite_deferred_;
    lite_deferred_ = false;
    SpecPush U{};
    if (push) {
        U.x = xc_; U.g = g_; U.a = lite_a_; U.a_s = lite_as_; U.M = lite_M_; U.S = lite_S_;
        U.sn = qn_S_.p + (size_t)lite_slot_ * ring_ld(n); U.yn = qn_Y_.p + (size_t)lite_slot_ * ring_ld(n);
        U.new_in_list = (count > 0 && slots[0] == lite_slot_) ? 1 : 0;   // (a pair with s·y ≤ 0 is written but does not join the history)
        push_counts_[0]++;
        qn_sgt_slot_ = -1;
    }
    // g, x, the ring / u — and with the state update: u_old / x, g, s, y, less the two reads of the pair formed in registers
    const double bytes = 8.0 * (double)n * (3.0 + hp + 2.0 * count + (push ? 5.0 - 2.0 * U.new_in_list : 0.0));
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LBFGS_FINAL)) return rc;
    switch (obj_->kind) {
    case CGO_OBJ_LSE: launch_spec<ObjLse>(big, push, grid, st, P, xc_, a_trial, Q, U); break;
    case CGO_OBJ_QUAD_DIAG: launch_spec<ObjQuadDiag>(big, push, grid, st, P, xc_, a_trial, Q, U); break;
    case CGO_OBJ_ROSENBROCK_PAIRED: launch_spec<ObjRosenPaired>(big, push, grid, st, P, xc_, a_trial, Q, U); break;
    case CGO_OBJ_USER: {   // the run-time compiled objective carries its own instantiations
        hipFunction_t f = obj_->rtc ? obj_->rtc->spec(big, push) : nullptr;
        if (!f) { set_error("internal: kernel missing from the run-time compiled objective module"); return CGO_EINVAL; }
        const double *xin = xc_;
        double at = a_trial;
        void *args[] = {(void *)&P, (void *)&xin, (void *)&at, (void *)&Q, (void *)&U};
        HIPCHK(hipModuleLaunchKernel(f, grid, 1, 1, BLOCK, 1, 1, 0, st, args, nullptr));
        break;
    }
    default: set_error("internal: no one-pass L-BFGS kernel for this objective"); return CGO_EINVAL;
    }
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_rows(ctx_, grid, NG)) return rc;
    double s[NG];
    if (int rc = fetch_sums(ctx_, s, MERGE_SUM, NG)) return rc;
    if (prof_on_) prof_commit(KK_LBFGS_FINAL, bytes);
    dir.gu = s[SP_GU]; dir.uu = s[SP_UU];
    if (!lse) {   // element-wise objective: the sums ARE the trial's and the next push's
        trial = Scal();
        trial.f = s[SE_F]; trial.gtu = s[SE_GTU]; trial.gtgt = s[SE_GTGT];
        std::memcpy(spec_s_, s, sizeof s);
        spec_a_ = a_trial; spec_count_ = count; spec_dphi_ = trial.gtu;
        for (int j = 0; j < count; ++j) spec_slots_[j] = slots[j];
        spec_valid_ = true;
        spec_unmat_ = true;    // g⁺ of this trial exists nowhere in memory (materialize() evaluates it again if somebody needs it)
        return CGO_OK;
    }
    // S' = Σ exp(xp − M_r) = exp(lse(xp) − lse(x)).  A first trial far out (overflow, or everything underflowing) is evaluated
    // the usual way instead — k_lse_stats works from the true maximum of xp — and nothing was speculated.
    const double Sp = s[SP_S];
    if (!(Sp >= 1e-280 && Sp <= 1e280) || !std::isfinite(s[SP_T])) {
        spec_refreshed_++;
        return lse_stats(0, 0, 0, a_trial, trial, false);
    }
    trial = Scal();
    trial.f = (Mr + std::log(Sp)) + 0.5 * obj_->s0 * s[SP_Q];   // ϕ = lse + ½λ‖xp‖², dϕ = softmax·u + λ xp·u (as lse_stats, reference M_r)
    trial.gtu = s[SP_T] / Sp + obj_->s0 * s[SP_R];
    std::memcpy(spec_s_, s, sizeof s);
    spec_Mr_ = Mr; spec_Sr_ = Sr; spec_a_ = a_trial; spec_count_ = count; spec_dphi_ = trial.gtu;
    for (int j = 0; j < count; ++j) spec_slots_[j] = slots[j];
    spec_valid_ = true;
    lse_a_ = a_trial; lse_M_ = Mr; lse_S_ = Sp;   // (M_r, S') describe xp as well as its own (max, Σ) would
    lse_have_ = true;
    return CGO_OK;
}

// The push for the step a_x from the sums the direction pass left, if a_x IS the step it speculated on: fills G (inner
// products with y-based entries: s_j·y, y_j·y, y_j·s — the caller adds its stored s_j·g, y_j·g), launches nothing; the
// state update itself (k_lbfgs_push_lite, or the next direction pass) is lbfgs_push_commit().  false = not available: take the usual path.
bool HipBackend::lbfgs_push_spec(double a_x, double a_s, int slot, const int *prev, int count, GramOut &G) {
    if (!spec_valid_ || push_pending_ || push_lite_pending_) return false;
    const bool lse = obj_->two_phase();
    if (std::memcmp(&a_x, &spec_a_, sizeof(double)) != 0 || (lse && std::memcmp(&a_x, &lse_a_, sizeof(double)) != 0)) return false;
    if (count != spec_count_) return false;
    for (int j = 0; j < count; ++j) if (prev[j] != spec_slots_[j]) return false;
    const double *s = spec_s_;
    if (!lse) {   // (spec_valid_ implies that no trial has been evaluated since the direction pass: trial() clears it)
        G.sy = a_s * s[SE_UY]; G.yy = s[SE_YY]; G.sgn = a_s * s[SE_GTU]; G.ygn = s[SE_YGT]; G.gtgt = s[SE_GTGT];
        bool ok = std::isfinite(G.sy) && std::isfinite(G.yy) && std::isfinite(G.sgn) && std::isfinite(G.ygn) && G.gtgt >= 1e-280 && G.gtgt <= 1e300;
        for (int j = 0; j < count; ++j) {
            const double *q = s + SP_PAIR + 5 * j;
            G.sjg[j] = q[0]; G.yjg[j] = q[1]; G.sjyn[j] = q[2]; G.yjyn[j] = q[3]; G.yjsn[j] = a_s * q[4];
            ok = ok && std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]) && std::isfinite(q[3]) && std::isfinite(q[4]);
        }
        if (!ok) return false;
        G.materialized = true; G.y_based = false;
        push_lite_pending_ = true;
        lite_a_ = a_x; lite_as_ = a_s; lite_slot_ = slot; lite_M_ = 0.0; lite_S_ = 1.0;
        spec_valid_ = false; spec_unmat_ = false;
        return true;
    }
    const double Sp = s[SP_S], lam = obj_->s0;
    const double kappa = spec_Sr_ / Sp, d = kappa - 1.0;                     // g⁺ = κ·p + λ·xp
    const double sup = s[SP_T] / spec_Sr_;                                    // Σ u·p
    const double E0 = s[SP_E0], E1 = s[SP_E0 + 1], E2 = s[SP_E0 + 2], E3 = s[SP_E0 + 3], E4 = s[SP_E0 + 4], E5 = s[SP_E0 + 5];
    G.sy = a_s * (E3 + d * sup);
    G.yy = E0 + 2.0 * d * E1 + d * d * E2;
    G.sgn = a_s * spec_dphi_;                                                 // s·g⁺ = a_s·(u·g⁺)
    G.ygn = kappa * E1 + lam * E5 + d * (kappa * E2 + lam * E4);
    // ‖g⁺‖², g⁺ = ĝ + d·p: from the element-wise small ĝ = p + λ·xp, not from Σp², Σp·xp, Σxp² (which cancel (‖p‖/‖g⁺‖)²-fold near a minimiser)
    G.gtgt = s[SP_GH2] + 2.0 * d * s[SP_GHP] + d * d * E2;
    bool ok = std::isfinite(kappa) && std::isfinite(G.sy) && std::isfinite(G.yy) && std::isfinite(G.ygn) &&
              G.gtgt >= 1e-280 && G.gtgt <= 1e300;   // (outside: the scaled-norm rare path wants a stored g⁺ — usual push)
    // y = ŷ + (κ − 1)·p is a sum of like-sized terms only while p does not dwarf y: p = S'·softmax(xp), so a step along which the
    // log-sum-exp RISES by more than log 2 (the ridge term paying for it) would have ŷ ≈ p ≫ y and the sums cancel S'-fold —
    // found by the seeded sweep (λ = 1e-6, iterates around −500: S' = 1e17, every y-sum came out 0).  Such a trial is as good a
    // trial as any (ϕ and dϕ are plain sums), but its push is the usual one.
    ok = ok && Sp <= 2.0;
    for (int j = 0; j < count; ++j) {
        const double *q = s + SP_PAIR + 5 * j;
        G.sjyn[j] = q[0] + d * q[1];
        G.yjyn[j] = q[2] + d * q[3];
        G.yjsn[j] = a_s * q[4];
        G.sjg[j] = G.yjg[j] = 0.0;
        ok = ok && std::isfinite(G.sjyn[j]) && std::isfinite(G.yjyn[j]) && std::isfinite(G.yjsn[j]);
    }
    if (!ok) return false;
    G.materialized = true; G.y_based = true;
    push_lite_pending_ = true;
    lite_a_ = a_x; lite_as_ = a_s; lite_slot_ = slot; lite_M_ = spec_Mr_; lite_S_ = Sp;
    spec_valid_ = false;
    return true;
}

int HipBackend::flush_lite() {
    if (!lite_deferred_) return CGO_OK;
    lite_deferred_ = false;
    return lbfgs_push_lite();
}

int HipBackend::lbfgs_push_lite() {
    HIPCHK(hipSetDevice(ctx_->device));
    const int64_t n = obj_->n_local;
    double *sn = qn_S_.p + (size_t)lite_slot_ * ring_ld(n), *yn = qn_Y_.p + (size_t)lite_slot_ * ring_ld(n);
    const double bytes = 8.0 * (double)n * (7.0 + (obj_->uses_param() ? 1.0 : 0.0));
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(KK_LBFGS_PUSH)) return rc;
    switch (obj_->kind) {
    case CGO_OBJ_LSE: launch_lite<ObjLse>(big, grid, st, xc_, u_.p, g_, sn, yn, obj_->p0.p, n, lite_a_, lite_as_, lite_M_, lite_S_, obj_->s0); break;
    case CGO_OBJ_QUAD_DIAG: launch_lite<ObjQuadDiag>(big, grid, st, xc_, u_.p, g_, sn, yn, obj_->p0.p, n, lite_a_, lite_as_, lite_M_, lite_S_, obj_->s0); break;
    case CGO_OBJ_ROSENBROCK_PAIRED: launch_lite<ObjRosenPaired>(big, grid, st, xc_, u_.p, g_, sn, yn, obj_->p0.p, n, lite_a_, lite_as_, lite_M_, lite_S_, obj_->s0); break;
    case CGO_OBJ_USER: {
        hipFunction_t f = obj_->rtc ? obj_->rtc->lite(big) : nullptr;
        if (!f) { set_error("internal: kernel missing from the run-time compiled objective module"); return CGO_EINVAL; }
        double *xa = xc_, *ga = g_, *sna = sn, *yna = yn;
        const double *ua = u_.p, *pa = obj_->p0.p;
        long long nn = n;
        double a = lite_a_, as = lite_as_, M = lite_M_, S = lite_S_, lam = obj_->s0;
        void *args[] = {&xa, &ua, &ga, &sna, &yna, &pa, &nn, &a, &as, &M, &S, &lam};
        HIPCHK(hipModuleLaunchKernel(f, grid, 1, 1, BLOCK, 1, 1, 0, st, args, nullptr));
        break;
    }
    default: set_error("internal: no one-pass L-BFGS kernel for this objective"); return CGO_EINVAL;
    }
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (prof_on_) prof_commit(KK_LBFGS_PUSH, bytes);
    qn_sgt_slot_ = -1;
    push_counts_[0]++;
    return CGO_OK;
}

// finalize + make the sums of the launch just enqueued available to the NEXT kernel on the
// device (dot_ptr) or, with a host communicator, on the host (dot_host).
int HipBackend::chain_sums(int grid, int slot, const double **dot_ptr, int *dot_count, double *dot_host) {
    hipStream_t st = ctx_->stream;
    if (int rc = finalize_rows(ctx_, grid, NS)) return rc;
    *dot_host = 0.0;
    if (ctx_->single()) { *dot_ptr = ctx_->out_dev; *dot_count = 1; return CGO_OK; }
    if (ctx_->shm()) {  // blocks live in host shared memory: one host round trip per step
        double sums[NS];
        if (int rc = fetch_sums(ctx_, sums)) return rc;
        *dot_ptr = nullptr; *dot_count = 0; *dot_host = sums[slot];
        return CGO_OK;
    }
    if (int rc = ctx_->ensure_gather()) return rc;
    const int dr = ctx_->comm->allgather_device(ctx_->out_dev, ctx_->gather_dev, NS, (void *)st);
    if (dr == 0) { *dot_ptr = ctx_->gather_dev; *dot_count = ctx_->world(); return CGO_OK; }
    if (dr > 0) return CGO_ECOMM;
    double sums[NS];
    if (int rc = fetch_sums(ctx_, sums)) return rc;  // host communicator: one round trip per step
    *dot_ptr = nullptr; *dot_count = 0; *dot_host = sums[slot];
    return CGO_OK;
}

int HipBackend::lbfgs_push(double a_x, double a_s, int slot, double &sy, double &yy) {
    HIPCHK(hipSetDevice(ctx_->device));
    const int64_t n = obj_->n_local;
    PushParams P;
    P.x = xc_; P.u = u_.p; P.g = g_; P.gt = gt_;
    P.s = qn_S_.p + (size_t)slot * ring_ld(n); P.y = qn_Y_.p + (size_t)slot * ring_ld(n);
    P.n = n; P.a = a_x; P.a_s = a_s; P.partials = ctx_->partials;
    const double bytes = 8.0 * (double)n * 7.0;
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    if (int rc = prof_begin(KK_LBFGS_PUSH)) return rc;
    if (big) k_lbfgs_push<true><<<grid, BLOCK, 0, ctx_->stream>>>(P);
    else k_lbfgs_push<false><<<grid, BLOCK, 0, ctx_->stream>>>(P);
    HIPCHK(hipGetLastError());
    if (int rc = prof_end()) return rc;
    total_launches_++;
    if (int rc = finalize_rows(ctx_, grid, NS)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s)) return rc;
    if (prof_on_) prof_commit(KK_LBFGS_PUSH, bytes);
    sy = s[PS_SY]; yy = s[PS_YY];
    qn_sgt_ = s[PS_SGT];
    qn_sgt_slot_ = slot;      // Σ s_slot·g⁺ is the first dot of the two-loop if this pair is kept
    std::swap(g_, gt_);       // g ← g⁺
    return CGO_OK;
}

int HipBackend::lbfgs_direction(const int *slots, const double *rho, int count, double gamma, Scal &out) {
    HIPCHK(hipSetDevice(ctx_->device));
    if (count == 0) return reset_dir(out);  // no curvature pairs yet: u = −g
    const int64_t n = obj_->n_local;
    const double bytes = 8.0 * (double)n * 4.0;
    const bool big = bytes > big_bytes();
    const int grid = big ? GRID_BIG : grid_for(n);
    hipStream_t st = ctx_->stream;
    auto S = [&](int slot) { return qn_S_.p + (size_t)slot * ring_ld(n); };
    auto Y = [&](int slot) { return qn_Y_.p + (size_t)slot * ring_ld(n); };
    LoopParams P;
    std::memset(&P, 0, sizeof(P));
    P.n = n; P.partials = ctx_->partials; P.alpha = qn_alpha_dev_; P.dot_stride = NS; P.dot_slot = S_GU;
    auto launch = [&](int kk, double nvec) -> int {
        if (int rc = prof_begin(kk)) return rc;
        if (big) k_lbfgs_loop<true><<<grid, BLOCK, 0, st>>>(P);
        else k_lbfgs_loop<false><<<grid, BLOCK, 0, st>>>(P);
        HIPCHK(hipGetLastError());
        if (int rc = prof_end()) return rc;
        total_launches_++;
        if (prof_on_) prof_commit(kk, 8.0 * (double)n * nvec);
        return CGO_OK;
    };
    // first dot  s_newest · g : already reduced by the push of this very pair, else one dot-only launch
    if (slots[0] == qn_sgt_slot_) {
        P.dot_ptr = nullptr; P.dot_count = 0; P.dot_host = qn_sgt_;
    } else {
        P.mode = 2; P.qin = g_; P.qout = u_.p; P.v = g_; P.w = S(slots[0]);
        if (int rc = launch(KK_LBFGS_LOOP, 2.0)) return rc;
        if (int rc = chain_sums(grid, S_GU, &P.dot_ptr, &P.dot_count, &P.dot_host)) return rc;
    }
    qn_sgt_slot_ = -1;
    for (int k = 0; k < count; ++k) {  // newest → oldest
        P.mode = 0; P.k = k; P.rho = rho[slots[k]];
        P.qin = (k == 0) ? g_ : u_.p; P.qout = u_.p; P.v = Y(slots[k]);
        P.apply_scale = (k == count - 1); P.scale = gamma; P.final_step = 0;
        P.w = (k < count - 1) ? S(slots[k + 1]) : Y(slots[count - 1]);
        if (int rc = launch(KK_LBFGS_LOOP, 4.0)) return rc;
        if (int rc = chain_sums(grid, S_GU, &P.dot_ptr, &P.dot_count, &P.dot_host)) return rc;
    }
    for (int k = count - 1; k >= 0; --k) {  // oldest → newest
        P.mode = 1; P.k = k; P.rho = rho[slots[k]];
        P.qin = u_.p; P.qout = u_.p; P.v = S(slots[k]);
        P.apply_scale = 0; P.final_step = (k == 0);
        P.w = (k > 0) ? Y(slots[k - 1]) : g_;
        if (int rc = launch(k == 0 ? KK_LBFGS_FINAL : KK_LBFGS_LOOP, 4.0)) return rc;
        if (k > 0) {
            if (int rc = chain_sums(grid, S_GU, &P.dot_ptr, &P.dot_count, &P.dot_host)) return rc;
        }
    }
    if (int rc = finalize_rows(ctx_, grid, NS)) return rc;
    double s[NS];
    if (int rc = fetch_sums(ctx_, s)) return rc;
    out.gu = s[S_GU]; out.uu = s[S_UU];
    return CGO_OK;
}

}  // namespace cgo
