// cgo_backend_cg.hip — the gradient-free multi-point CG family on the device: k_cg / k_chain launches and their reduction tails,
// the on-device line-search controller (armed rounds), the resident solver.
#include "cgo_backend_internal.hpp"

#include "cgo_kernels.hip.hpp"
#include "cgo_kernels_cg.hip.hpp"
#include "cgo_kernels_chain.hip.hpp"
#include "cgo_kernels_resident.hip.hpp"

namespace cgo {

using namespace dev;

// ---- gradient-free multi-point CG family (cgo_kernels_cg.hip.hpp) ---------------------------
double bytes_r(int obj_kind, int mode, int64_t n, bool has_param) {
    const int p = (obj_kind == CGO_OBJ_QUAD_DIAG || has_param) ? 1 : 0;
    int v = 0;
    if (mode == R_INIT) v = 1 + p + 1;
    else if (mode == R_TRIAL) v = 2 + p;
    else if (mode == (R_ACCEPT | R_DIR | R_TRIAL)) v = 2 + p + 2;
    else if (mode == (R_ACCEPT | R_DIR)) v = 2 + p + 2;
    else if (mode == R_ACCEPT) v = 2 + 1;
    else if (mode == R_RESET) v = 1 + p + 1;
    else if (mode == R_UPG) v = 2 + p;
    else if (mode == R_GRAD) v = 1 + p + 1;
    else if (mode == R_GRADT) v = 2 + p + 1;
    else if (mode == R_DIR || mode == (R_DIR | R_TRIAL)) v = 2 + p + 1;
    else if (mode == R_PROJ) v = 3 + p + 1;
    else if (mode == R_EDGES) v = 0;
    return 8.0 * (double)n * (double)v;
}

template <class Obj, bool BIG>
static int launch_cg(int mode, int npts, const RParams &P, int grid, hipStream_t st) {
    switch (mode) {
    case R_INIT: k_cg<Obj, R_INIT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_TRIAL:
        if (npts == 1) k_cg<Obj, R_TRIAL, 1, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 3) k_cg<Obj, R_TRIAL, 3, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 5) k_cg<Obj, R_TRIAL, 5, BIG><<<grid, BLOCK, 0, st>>>(P);
        else k_cg<Obj, R_TRIAL, 7, BIG><<<grid, BLOCK, 0, st>>>(P);
        break;
    case R_ACCEPT | R_DIR | R_TRIAL:
        if (P.tail.ctl) {   // a whole controller round in this launch (never BIG: pipe_fused)
            if (npts == 1) k_cg_armed<Obj, 1><<<grid, BLOCK, 0, st>>>(P);
            else if (npts == 3) k_cg_armed<Obj, 3><<<grid, BLOCK, 0, st>>>(P);
            else if (npts == 5) k_cg_armed<Obj, 5><<<grid, BLOCK, 0, st>>>(P);
            else k_cg_armed<Obj, 7><<<grid, BLOCK, 0, st>>>(P);
            break;
        }
        if (npts == 1) k_cg<Obj, R_ACCEPT | R_DIR | R_TRIAL, 1, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 3) k_cg<Obj, R_ACCEPT | R_DIR | R_TRIAL, 3, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 5) k_cg<Obj, R_ACCEPT | R_DIR | R_TRIAL, 5, BIG><<<grid, BLOCK, 0, st>>>(P);
        else k_cg<Obj, R_ACCEPT | R_DIR | R_TRIAL, 7, BIG><<<grid, BLOCK, 0, st>>>(P);
        break;
    case R_ACCEPT | R_DIR: k_cg<Obj, R_ACCEPT | R_DIR, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_ACCEPT: k_cg<Obj, R_ACCEPT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_RESET: k_cg<Obj, R_RESET, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_UPG: k_cg<Obj, R_UPG, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_GRAD: k_cg<Obj, R_GRAD, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_GRADT: k_cg<Obj, R_GRADT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_DIR: k_cg<Obj, R_DIR, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_DIR | R_TRIAL:
        if (npts == 1) k_cg<Obj, R_DIR | R_TRIAL, 1, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 3) k_cg<Obj, R_DIR | R_TRIAL, 3, BIG><<<grid, BLOCK, 0, st>>>(P);
        else if (npts == 5) k_cg<Obj, R_DIR | R_TRIAL, 5, BIG><<<grid, BLOCK, 0, st>>>(P);
        else k_cg<Obj, R_DIR | R_TRIAL, 7, BIG><<<grid, BLOCK, 0, st>>>(P);
        break;
    case R_PROJ: k_cg<Obj, R_PROJ, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    default: return -1;
    }
    return 0;
}

// Row width of a CG launch: 7 sums per trial point + 2 direction sums, padded (10 or 24).
static inline int rows_for(int npts) { return npts == 1 ? NR1 : (npts == 3 ? NR : (npts == 5 ? NR5 : NR7)); }
static inline int npts_for(int k) { return k <= 1 ? 1 : (k <= 3 ? 3 : (k <= 5 ? 5 : 7)); }  // kernel variant for k trial steps

// stencil launches carry one or three trial points: three only where the mode evaluates trials at all
static inline int chain_npts(int mode, int npts) { return ((mode & R_TRIAL) && npts >= 3) ? 3 : 1; }

int HipBackend::launch_r(int kk, int mode, double a_acc, double beta, const double *a, int k, bool fetch,
                         double *sums) {
    if (int rc = pipe_drain()) return rc;
    pipe_streak_ = 0;  // a host-driven launch: the streak of controller-eligible launches ends
    int grid = 0;
    const int npts = npts_for(k);
    if (int rc = launch_r_kernel(kk, mode, a_acc, beta, a, k, npts, nullptr, &grid)) return rc;
    total_launches_++;
    const bool has_sums = (mode != R_ACCEPT && mode != R_GRAD && mode != R_GRADT);
    const bool fused = has_sums && tail_fused(grid);   // the launch's last workgroup already left the sums (finish_tail)
    if (has_sums && chain()) {   // 24- or 32-slot rows: the sums + this rank's eight edge values (cgo_kernels_chain.hip.hpp)
        const bool three = chain_npts(mode, npts) == 3;
        const int W = three ? NRC3 : NRC, edge = three ? RC3_EDGE : RC_EDGE, nsums = three ? NR : NR1;
        if (!fused) { if (int rc = finalize_rows(ctx_, grid, W, true)) return rc; }
        const int Wd = ctx_->world(), me = ctx_->rank();
        std::vector<double> raw((size_t)W * Wd);
        double all[NRC3];
        if (int rc = fetch_sums(ctx_, all, MERGE_SUM, W, raw.data())) return rc;
        if (sums) std::memcpy(sums, all, sizeof(double) * nsums);
        if (me > 0) {           // left neighbour's LAST two elements
            const double *e = raw.data() + (size_t)(me - 1) * W + edge + 4;
            halo_xl_[0] = e[0]; halo_xl_[1] = e[1]; halo_ul_[0] = e[2]; halo_ul_[1] = e[3];
        }
        if (me < Wd - 1) {       // right neighbour's FIRST two elements
            const double *e = raw.data() + (size_t)(me + 1) * W + edge;
            halo_xr_[0] = e[0]; halo_xr_[1] = e[1]; halo_ur_[0] = e[2]; halo_ur_[1] = e[3];
        }
    } else if (has_sums) {
        if (!fused) { if (int rc = finalize_rows(ctx_, grid, rows_for(npts), true)) return rc; }
        if (fetch) {
            if (int rc = fetch_sums(ctx_, sums, MERGE_SUM, rows_for(npts))) return rc;
        }
    }
    if (prof_on_) prof_commit(kk, bytes_r(obj_->kind, mode, obj_->n_local, obj_->uses_param()));
    return CGO_OK;
}

// Fused reduction tail (finish_tail): a host-driven launch of the k_cg / k_chain family takes the next sequence number
// itself and publishes where a finalize launch would have.
// Only where the launch is short: at 4096 workgroups the ≈ 0.5 M slot and ticket atomics and the finisher's chain cost the
// pure-HBM launch what the two finalize launches did (n = 1e8: 671 → 683 µs, 1 236 vs 1 230 it/s; gpurun_out/r02_ft).
// A controller-armed round as ONE launch (tail_ctl): wherever the fused tail applies, except for run-time compiled
// objectives, whose kernels carry no controller code.
bool HipBackend::pipe_fused(int grid) const {
    return tail_fused(grid) && obj_->kind != CGO_OBJ_USER && !chain() && ctl_fused_;
}
bool HipBackend::tail_fused(int grid) const {
    static const int cap = [] { const char *e = getenv("CGO_FUSED_TAIL_MAX_GRID"); int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    return ctx_->fused_tail && grid <= cap && grid <= TAIL_GROUP * TAIL_GROUP;
}
Tail HipBackend::make_tail(bool on) {
    Tail t{};
    if (!on) return t;
    ctx_->seq++;
    t.partials2 = ctx_->partials2_f; t.tickets = ctx_->tickets; t.out = ctx_->out_dev;
    t.strict = ctx_->tail_strict ? 1 : 0;
    ctx_->pub_target(&t.host_out, &t.host_seq);
    ctx_->pub_checked = (t.host_out != nullptr) && !ctx_->tail_strict;
    t.seq = ctx_->seq;
    return t;
}

// the k_cg launch itself (bracketed by the profiling events); `ctl` non-null = controller-armed
int HipBackend::launch_r_kernel(int kk, int mode, double a_acc, double beta, const double *a, int k, int npts,
                                const CtlArgs *ctl, int *grid_out) {
    HIPCHK(hipSetDevice(ctx_->device));
    if (obj_->uses_param() && !obj_->p0_set) { set_error("objective parameter vector (slot 0) was never set"); return CGO_ESTATE; }
    const int64_t n = obj_->n_local;
    if (mode & (R_GRAD | R_GRADT)) { if (int rc = ensure_ga()) return rc; }
    const bool has_sums = (mode != R_ACCEPT && mode != R_GRAD && mode != R_GRADT);
    if (chain()) {
        const double bytes = bytes_r(obj_->kind, mode, n, false);
        const bool big = bytes > big_bytes(mode == R_TRIAL || mode == R_UPG);
        const int grid = big ? GRID_BIG : grid_cg(n, 1);
        *grid_out = grid;
        if (int rc = prof_begin(kk)) return rc;
        if (int rc = launch_chain_kernel(mode, a_acc, beta, a, k, chain_npts(mode, npts), big, grid, make_tail(has_sums && !ctl && tail_fused(grid)))) return rc;
        return prof_end();
    }
    RParams P;
    P.x = xc_; P.u = uc_; P.gout = ga_.p; P.p0 = obj_->p0.p; P.n = n;
    P.xo = xc_; P.uo = uc_;
    P.a_acc = a_acc; P.beta = beta; P.s0 = obj_->s0; P.partials = ctx_->partials;
    P.ctl = ctl;
    P.x2 = xn_;
    for (int j = 0; j < MAXP; ++j) P.a[j] = (a && j < k) ? a[j] : ((a && k > 0) ? a[k - 1] : 0.0);
    const double bytes = bytes_r(obj_->kind, mode, n, obj_->uses_param());
    const bool big = bytes > big_bytes(mode == R_TRIAL || mode == R_UPG);
    const int grid = big ? GRID_BIG : grid_cg(n, npts);
    *grid_out = grid;
    P.tail = make_tail(has_sums && !ctl && tail_fused(grid));
    if (ctl && pipe_fused(grid)) {
        P.tail.partials2 = ctx_->partials2_f; P.tail.tickets = ctx_->tickets; P.tail.out = ctx_->out_dev;
        P.tail.strict = ctx_->tail_strict ? 1 : 0;
        P.tail.ctl = ctl_dev_; P.tail.ctl_rec = ctl_rec_; P.tail.ctl_seq = ctl_seq_;
        if (!ctx_->single()) {   // the finisher exchanges its block with the peers' GPUs itself (tail_exchange)
            P.tail.xw = ctx_->world(); P.tail.xme = ctx_->rank(); P.tail.xseq0 = epoch_ << 40;
            for (int r = 0; r < P.tail.xw && r < 8; ++r) P.tail.xmail[r] = ctx_->comm->dev_mailbox(r);
        }
    }
    if (P.tail.tickets) P.partials = ctx_->partials_f;
    const bool wr_x = (mode & R_ACCEPT) != 0, wr_u = (mode & (R_DIR | R_INIT | R_RESET)) != 0;
    const bool pp = big && !ctl && (wr_x || wr_u) && !(mode & R_PROJ) && pingpong_ready();
    if (pp && wr_x) P.xo = xalt_;
    if (pp && wr_u) P.uo = ualt_;
    if (mode == R_PROJ && !xn_) { set_error("internal: no second iterate buffer"); return CGO_ESTATE; }
    hipStream_t st = ctx_->stream;
    if (int rc = prof_begin(kk)) return rc;
    int r = -2;
    switch (obj_->kind) {
    case CGO_OBJ_QUAD_DIAG: r = big ? launch_cg<ObjQuadDiag, true>(mode, npts, P, grid, st) : launch_cg<ObjQuadDiag, false>(mode, npts, P, grid, st); break;
    case CGO_OBJ_ROSENBROCK_PAIRED: r = big ? launch_cg<ObjRosenPaired, true>(mode, npts, P, grid, st) : launch_cg<ObjRosenPaired, false>(mode, npts, P, grid, st); break;
    case CGO_OBJ_BOOTH: r = big ? launch_cg<ObjBooth, true>(mode, npts, P, grid, st) : launch_cg<ObjBooth, false>(mode, npts, P, grid, st); break;
    case CGO_OBJ_USER:
        if (!obj_->rtc) { set_error("user objective has no compiled module"); return CGO_EINVAL; }
        if (int rc = launch_module(obj_->rtc->cg(mode, npts, big), &P, grid, st)) return rc;
        r = 0;
        break;
    default: break;
    }
    if (r) { set_error("internal: CG kernel mode not instantiated"); return CGO_EINVAL; }
    HIPCHK(hipGetLastError());
    if (pp && wr_x) std::swap(xc_, xalt_);
    if (pp && wr_u) std::swap(uc_, ualt_);
    return prof_end();
}

// ---- chained Rosenbrock: the stencil launches (cgo_kernels_chain.hip.hpp) ----------------------------------------
template <bool BIG>
static int launch_chain(int mode, int npts, const ChainParams &P, int grid, hipStream_t st) {
    switch (mode) {
    case R_INIT: k_chain<R_INIT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_TRIAL:
        if (npts == 3) k_chain<R_TRIAL, 3, BIG><<<grid, BLOCK, 0, st>>>(P);
        else k_chain<R_TRIAL, 1, BIG><<<grid, BLOCK, 0, st>>>(P);
        break;
    case R_ACCEPT | R_DIR | R_TRIAL:
        if (npts == 3) k_chain<R_ACCEPT | R_DIR | R_TRIAL, 3, BIG><<<grid, BLOCK, 0, st>>>(P);
        else k_chain<R_ACCEPT | R_DIR | R_TRIAL, 1, BIG><<<grid, BLOCK, 0, st>>>(P);
        break;
    case R_ACCEPT | R_DIR: k_chain<R_ACCEPT | R_DIR, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_ACCEPT: k_chain<R_ACCEPT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_RESET: k_chain<R_RESET, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_UPG: k_chain<R_UPG, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_GRAD: k_chain<R_GRAD, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_GRADT: k_chain<R_GRADT, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    case R_EDGES: k_chain<R_EDGES, 1, BIG><<<grid, BLOCK, 0, st>>>(P); break;
    default: return -1;
    }
    return 0;
}

int HipBackend::launch_chain_kernel(int mode, double a_acc, double beta, const double *a, int k, int npts, bool big, int grid, const Tail &tail) {
    ChainParams P;
    P.tail = tail;
    for (int j = 0; j < 3; ++j) P.a[j] = (a && j < k) ? a[j] : ((a && k > 0) ? a[k - 1] : 0.0);
    P.x = xc_; P.u = uc_; P.xo = xc_; P.uo = uc_; P.gout = ga_.p;
    P.odd = (int)(obj_->n_local & 1);
    P.n = obj_->n_local + P.odd; P.a_acc = a_acc; P.beta = beta; P.partials = tail.tickets ? ctx_->partials_f : ctx_->partials;
    for (int j = 0; j < 2; ++j) { P.hxl[j] = halo_xl_[j]; P.hul[j] = halo_ul_[j]; P.hxr[j] = halo_xr_[j]; P.hur[j] = halo_ur_[j]; }
    // the global vector ends where this rank's shard touches its ends
    P.has_left = obj_->offset > 0 ? 1 : 0;
    P.has_right = obj_->offset + obj_->n_local < obj_->n_global ? 1 : 0;
    const bool wr_x = (mode & R_ACCEPT) != 0, wr_u = (mode & (R_DIR | R_INIT | R_RESET)) != 0;
    if (wr_x) P.xo = xalt_;
    if (wr_u) P.uo = ualt_;
    const int r = big ? launch_chain<true>(mode, npts, P, grid, ctx_->stream) : launch_chain<false>(mode, npts, P, grid, ctx_->stream);
    if (r) { set_error("internal: chain kernel mode not instantiated"); return CGO_EINVAL; }
    HIPCHK(hipGetLastError());
    if (wr_x) std::swap(xc_, xalt_);
    if (wr_u) std::swap(uc_, ualt_);
    return CGO_OK;
}

// The instantiation a launch of kind `kk` uses under the current policy, as rocprofv3 prints it minus namespaces.
std::string HipBackend::kernel_symbol(int kk) const {
    const char *on = obj_->kind == CGO_OBJ_QUAD_DIAG ? "ObjQuadDiag" : obj_->kind == CGO_OBJ_ROSENBROCK_PAIRED ? "ObjRosenPaired"
                     : obj_->kind == CGO_OBJ_BOOTH ? "ObjBooth" : obj_->kind == CGO_OBJ_USER ? "UserObjective" : "";
    const int64_t n = obj_->n_local;
    const bool hp = obj_->uses_param();
    char buf[160];
    if (rmode_) {
        int mode = -1, npts = 1;
        switch (kk) {
        case KK_INIT: mode = R_INIT; break;
        case KK_TRIAL: mode = R_TRIAL; npts = npts_for(std::min(max_points(), 3)); break;
        case KK_ACCEPT_DIR_TRIAL: mode = R_ACCEPT | R_DIR | R_TRIAL; npts = npts_for(max_points()); break;
        case KK_ACCEPT_DIR: mode = R_ACCEPT | R_DIR; break;
        case KK_ACCEPT_ONLY: mode = R_ACCEPT; break;
        case KK_RESET_DIR: mode = R_RESET; break;
        case KK_UPG_NORM: mode = R_UPG; break;
        case KK_DIR_TRIAL: mode = R_DIR | R_TRIAL; npts = npts_for(max_points()); break;
        case KK_SYS_PROJECT: mode = R_PROJ; break;
        default: return "";
        }
        const bool big = bytes_r(obj_->kind, mode, n, hp) > big_bytes(mode == R_TRIAL || mode == R_UPG);
        if (chain()) snprintf(buf, sizeof buf, "k_chain<%d, %d, %s>", mode, chain_npts(mode, npts), big ? "true" : "false");
        else snprintf(buf, sizeof buf, "k_cg<%s, %d, %d, %s>", on, mode, npts, big ? "true" : "false");
        return buf;
    }
    if (obj_->two_phase()) {
        if (kk == KK_LSE_STATS) return "k_lse_stats";
        if (kk == KK_LSE_GRAD) return "k_lse_grad";
        const bool big_ring = 8.0 * (double)n * (3.0 + 2.0 * std::max(qn_m_ - 1, 0)) > big_bytes();
        if (kk == KK_LBFGS_FINAL && qn_m_ > 0) {   // the L-BFGS passes of the log-sum-exp objective (a full ring assumed for the policy bit)
            if (spec_on_ && qn_m_ - 1 <= SPEC_MAXC) { snprintf(buf, sizeof buf, "k_lbfgs_combine_spec<ObjLse, %s, %s>", big_ring ? "true" : "false", spec_fuse_push_ ? "true" : "false"); return buf; }
            if (gram_on_) { snprintf(buf, sizeof buf, "k_lbfgs_combine_lse<%s>", big_ring ? "true" : "false"); return buf; }
            return "k_lbfgs_loop";
        }
        if (kk == KK_LBFGS_PUSH && qn_m_ > 0) {
            if (spec_on_ && qn_m_ - 1 <= SPEC_MAXC && !spec_fuse_push_) { snprintf(buf, sizeof buf, "k_lbfgs_push_lite<ObjLse, %s>", 8.0 * (double)n * 7.0 > big_bytes() ? "true" : "false"); return buf; }
            if (gram_on_) { snprintf(buf, sizeof buf, fuse_grad_ && x2_.p ? "k_lbfgs_push_gram_lse<%s>" : "k_lbfgs_push_gram<%s>", big_ring ? "true" : "false"); return buf; }
            return "k_lbfgs_push";
        }
        return "";
    }
    int mode = -1;
    switch (kk) {
    case KK_INIT: mode = M_INIT; break;
    case KK_TRIAL: mode = need_beta_ ? (M_TRIAL | M_BETA) : M_TRIAL; break;
    case KK_ACCEPT_DIR_TRIAL: mode = M_ACCEPT | M_DIR | M_TRIAL | M_BETA; break;
    case KK_ACCEPT_DIR: mode = M_ACCEPT | M_DIR; break;
    case KK_ACCEPT_ONLY: mode = M_ACCEPT; break;
    case KK_RESET_DIR: mode = M_RESET; break;
    case KK_UPG_NORM: mode = M_UPG; break;
    case KK_LBFGS_PUSH: return gram_on_ ? "k_lbfgs_push_gram" : "k_lbfgs_push";
    case KK_LBFGS_LOOP: return "k_lbfgs_loop";
    case KK_LBFGS_FINAL:
        if (spec_on_ && qn_m_ > 0 && qn_m_ - 1 <= SPEC_MAXC) {   // the one-pass form (a full ring assumed for the policy bit)
            snprintf(buf, sizeof buf, "k_lbfgs_combine_spec<%s, %s, %s>", on, 8.0 * (double)n * (3.0 + (hp ? 1.0 : 0.0) + 2.0 * (qn_m_ - 1)) > big_bytes() ? "true" : "false",
                     spec_fuse_push_ ? "true" : "false");
            return buf;
        }
        return gram_on_ ? "k_lbfgs_combine" : "k_lbfgs_loop";
    default: return "";
    }
    const bool objective_mode = (mode & (M_TRIAL | M_INIT)) != 0;
    snprintf(buf, sizeof buf, "k_fused<%s, %d, %s>", objective_mode ? on : "ObjQuadDiag", mode, is_big(obj_->kind, mode, n, hp, pol_.hbm_stream_bytes) ? "true" : "false");
    return buf;
}

// ---- on-device controller (cgo_ctl.hpp) ------------------------------------------------------
// Device block: the controller's config and state, and the argument block the armed launches read.
// `round` numbers the rounds of a solve on the DEVICE: the reduce/controller kernel derives its record slot and its
// sequence word from it, so that a round's kernels carry no per-round host argument at all and whole batches of
// rounds replay from one instantiated hipGraph (pipe_launch_graph).
// (struct CtlDev: cgo_kernels_cg.hip.hpp — the armed launches' own finisher reads and writes it too)

__global__ void k_ctl_init(CtlDev *d, const CtlConfig cfg, const CtlState st, unsigned long long round) {
    d->cfg = cfg;
    d->st = st;
    d->round = round;
    CtlArgs a;
    a.a_acc = st.a_acc; a.beta = st.beta; a.go = st.go;
    for (int j = 0; j < CTL_MAXP; ++j) a.a[j] = st.a[j];
    d->args = a;
}

// Final reduction stage of a controller-armed launch + the controller itself: rows → sums →
// ctl_step() → arguments of the next launch (device memory) and the round's record (pinned host
// memory, released with a sequence word the host polls).
// One lane running scalar code is the slow part of this kernel (a dependent global load costs ≈ 1–2 µs, a
// PCIe store ≈ 0.2 µs): the device block is staged into LDS and the results are written back — state and
// arguments to HBM, the 30-word record to pinned host memory — by as many lanes as there are words.


template <int N, int THREADS>
__global__ __launch_bounds__(THREADS) void k_finalize_ctl(const double *partials, int rows, double *out, CtlDev *d,
                                                          CtlRecord *rec_ring, unsigned long long *seq_ring) {
    constexpr int G = BLOCK / N;   // the k_cg family's summation order (finalize_rows canon, finish_tail)
    constexpr int WD = sizeof(CtlDev) / 8, WR = sizeof(CtlRecord) / 8;
    __shared__ double sm[G][N];
    __shared__ double fin[CTL_NSUMS];
    __shared__ CtlDev sd;
    __shared__ CtlRecord sr;
    const int tid = threadIdx.x;
    if (tid < WD) ((unsigned long long *)&sd)[tid] = ((const unsigned long long *)d)[tid];
    if (tid < CTL_NSUMS) fin[tid] = 0.0;
    __syncthreads();
    const bool go = sd.st.go != 0;
    const unsigned long long round = sd.round, seq = round + 1;
    CtlRecord *rec_host = rec_ring + (round % PIPE_RING);
    unsigned long long *seq_host = seq_ring + (round % PIPE_RING);
    if (go) {  // same summation order as k_finalize_t: the record must hold what a host-driven launch would
        if (tid < G * N) {
            double t = 0.0;
            const long long total = (long long)rows * N;
            for (long long i = tid; i < total; i += G * N) t += partials[i];
            sm[tid / N][tid % N] = t;
        }
        __syncthreads();
        if (tid < N) {
            double v = 0.0;
#pragma unroll
            for (int g = 0; g < G; ++g) v += sm[g][tid];
            out[tid] = v;
            fin[tid] = v;
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (go) {
            ctl_step(sd.cfg, sd.st, fin, sr);
            CtlArgs a;
            a.a_acc = sd.st.a_acc; a.beta = sd.st.beta; a.go = sd.st.go;
            for (int j = 0; j < CTL_MAXP; ++j) a.a[j] = sd.st.a[j];
            sd.args = a;
        } else {
            for (int i = 0; i < CTL_NSUMS; ++i) sr.sums[i] = 0.0;
            sr.a_acc = 0.0; sr.beta = 0.0;
            for (int j = 0; j < CTL_MAXP; ++j) sr.a[j] = 0.0;
            sr.npts = -1; sr.accepted = 0; sr.xwait = 0;
        }
        sd.round = round + 1;
    }
    __syncthreads();
    if (go && tid < WD) ((unsigned long long *)d)[tid] = ((const unsigned long long *)&sd)[tid];
    if (!go && tid == 0) d->round = round + 1;
    if (tid < WR) {
        ((unsigned long long *)rec_host)[tid] = ((const unsigned long long *)&sr)[tid];
        __threadfence_system();
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(seq_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Armed rounds: one rank — or several whose GPUs exchange their blocks themselves (device mailboxes, cgo_comm.hip), which only
// the single-launch form of a round does (tail_ctl): built-in objective, grid-stride launch with a fused tail.
int HipBackend::ctl_depth() const {
    if (!(rmode_ && ctx_->host_publish && !obj_->two_phase())) return 0;
    if (ctx_->single()) return ctl_depth_;
    if (!ctx_->dev_exchange() || ctx_->force_gather) return 0;
    const bool big = bytes_r(obj_->kind, R_ACCEPT | R_DIR | R_TRIAL, obj_->n_local, obj_->uses_param()) > big_bytes(false);
    return (!big && pipe_fused(grid_cg(obj_->n_local, policy_points()))) ? ctl_depth_ : 0;
}

int HipBackend::pipe_alloc() {
    if (ctl_dev_) return CGO_OK;
    HIPCHK(hipMalloc(&ctl_dev_, sizeof(CtlDev)));
    HIPCHK(hipHostMalloc((void **)&ctl_rec_, sizeof(CtlRecord) * PIPE_RING, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&ctl_seq_, sizeof(unsigned long long) * PIPE_RING, hipHostMallocDefault));
    std::memset(ctl_rec_, 0, sizeof(CtlRecord) * PIPE_RING);
    std::memset(ctl_seq_, 0, sizeof(unsigned long long) * PIPE_RING);
    pipe_prof_.assign(PIPE_RING, {-1, 0u});
    return CGO_OK;
}

// At solver creation: the controller's blocks, and ONE armed round with the controller stopped — a no-op that files an idle
// record — so that the first launches of k_ctl_init and of the armed kernel (≈ 60 µs each of code-object set-up) do not
// fall into the first armed iteration (BASELINE config 1 runs 25 iterations in all: 14.4k vs 16.5k it/s).
int HipBackend::prepare_controller() {
    if (ctl_depth() <= 0) return CGO_OK;
    if (int rc = pipe_alloc()) return rc;
    if (obj_->uses_param() && !obj_->p0_set) return CGO_OK;   // nothing to launch on yet
    HIPCHK(hipSetDevice(ctx_->device));
    CtlConfig cc{};
    CtlState st{};
    st.go = 0;
    pipe_npts_ = max_points();
    k_ctl_init<<<1, 1, 0, ctx_->stream>>>((CtlDev *)ctl_dev_, cc, st, pipe_enq_);
    HIPCHK(hipGetLastError());
    if (int rc = pipe_enqueue_round()) return rc;
    return pipe_drain();
}

// the kernels of one controller-armed round: k_cg reading its scalars from the device block, then reduce + controller.
// No argument depends on the round (record slot and sequence number come from CtlDev::round), so the same launches can
// be captured into a hipGraph.
int HipBackend::pipe_round_kernels() {
    int grid = 0;
    const int npts = pipe_npts_, ns = rows_for(npts);
    CtlDev *d = (CtlDev *)ctl_dev_;
    if (int rc = launch_r_kernel(KK_ACCEPT_DIR_TRIAL, R_ACCEPT | R_DIR | R_TRIAL, 0.0, 0.0, nullptr, 0, npts, &d->args, &grid)) return rc;
    pipe_checked_ = pipe_fused(grid) && !ctx_->tail_strict;
    if (pipe_fused(grid)) return CGO_OK;   // the launch's own finisher reduced, ran the controller and published the record
    hipStream_t st = ctx_->stream;
    const double *src = ctx_->partials;
    int nrows = grid;
    if (grid > TAIL_GROUP) {
        const int nb = (grid + TAIL_GROUP - 1) / TAIL_GROUP;
        if (ns == NR) k_finalize_t<NR, BLOCK><<<nb, BLOCK, 0, st>>>(ctx_->partials, TAIL_GROUP, grid, ctx_->partials2, nullptr, nullptr, 0);
        else if (ns == NR5) k_finalize_t<NR5, BLOCK><<<nb, BLOCK, 0, st>>>(ctx_->partials, TAIL_GROUP, grid, ctx_->partials2, nullptr, nullptr, 0);
        else if (ns == NR7) k_finalize_t<NR7, BLOCK><<<nb, BLOCK, 0, st>>>(ctx_->partials, TAIL_GROUP, grid, ctx_->partials2, nullptr, nullptr, 0);
        else k_finalize_t<NS, BLOCK><<<nb, BLOCK, 0, st>>>(ctx_->partials, TAIL_GROUP, grid, ctx_->partials2, nullptr, nullptr, 0);
        HIPCHK(hipGetLastError());
        src = ctx_->partials2;
        nrows = nb;
    }
    CtlRecord *rec = (CtlRecord *)ctl_rec_;
    if (ns == NR) k_finalize_ctl<NR, 768><<<1, 768, 0, st>>>(src, nrows, ctx_->out_dev, d, rec, ctl_seq_);
    else if (ns == NR5) k_finalize_ctl<NR5, 768><<<1, 768, 0, st>>>(src, nrows, ctx_->out_dev, d, rec, ctl_seq_);
    else if (ns == NR7) k_finalize_ctl<NR7, 768><<<1, 768, 0, st>>>(src, nrows, ctx_->out_dev, d, rec, ctl_seq_);
    else k_finalize_ctl<NS, BLOCK><<<1, BLOCK, 0, st>>>(src, nrows, ctx_->out_dev, d, rec, ctl_seq_);
    HIPCHK(hipGetLastError());
    return CGO_OK;
}

// one round, launched kernel by kernel (with a HIP-event sample when the profiler picks it)
int HipBackend::pipe_enqueue_round() {
    if (int rc = pipe_round_kernels()) return rc;
    const int idx = (int)(pipe_enq_ % PIPE_RING);
    pipe_prof_[idx] = {prof_cur_ ? ring_used_ - 1 : -1, prof_gen_};
    prof_cur_ = false;
    pipe_enq_++;
    return CGO_OK;
}

// `rounds` rounds as ONE hipGraphLaunch: the per-launch host cost (≈ 3.5 µs per kernel, two or three kernels per round)
// is what kept the device waiting for the host at small n although the controller needs no host decision
// (DESIGN.md §2.7).  Instantiated once per (rounds, row width, buffers) and replayed.
int HipBackend::pipe_launch_graph(int rounds) {
    HIPCHK(hipSetDevice(ctx_->device));
    PipeGraph *g = nullptr;
    for (auto &c : graphs_)
        if (c.rounds == rounds && c.npts == pipe_npts_ && c.x == xc_ && c.u == uc_ && c.p0 == obj_->p0.p && c.n == obj_->n_local) { g = &c; break; }
    if (!g) {
        hipStream_t st = ctx_->stream;
        hipGraph_t graph = nullptr;
        capturing_ = true;
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        int rc = CGO_OK;
        if (e == hipSuccess) {
            for (int r = 0; r < rounds && rc == CGO_OK; ++r) rc = pipe_round_kernels();
            hipError_t e2 = hipStreamEndCapture(st, &graph);
            if (e2 != hipSuccess) e = e2;
        }
        capturing_ = false;
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || !graph) { set_error(std::string("hipGraph capture of controller rounds failed: ") + hipGetErrorString(e)); (void)hipGetLastError(); return CGO_EHIP; }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { set_error(std::string("hipGraphInstantiate failed: ") + hipGetErrorString(e)); return CGO_EHIP; }
        graphs_.push_back(PipeGraph{exec, rounds, pipe_npts_, xc_, uc_, obj_->p0.p, obj_->n_local});
        g = &graphs_.back();
    }
    HIPCHK(hipGraphLaunch((hipGraphExec_t)g->exec, ctx_->stream));
    for (int r = 0; r < rounds; ++r) {
        pipe_prof_[(int)(pipe_enq_ % PIPE_RING)] = {-1, prof_gen_};
        pipe_enq_++;
    }
    graph_rounds_ += rounds;
    return CGO_OK;
}

// enqueue `count` more rounds: graphs of 8 / 4 / 2 rounds where possible.  With the profiler on, every 4th batch goes
// kernel by kernel so that the HIP-event samples of the armed launches keep coming.
int HipBackend::pipe_enqueue(int64_t count) {
    const bool eager = !graph_on_ || (prof_on_ && ((pipe_batches_++ & 3) == 0));
    while (count > 0) {
        int c = 1;
        if (!eager) { c = 8; while (c > count) c >>= 1; }
        if (c == 1) { if (int rc = pipe_enqueue_round()) return rc; }
        else if (int rc = pipe_launch_graph(c)) return rc;
        count -= c;
    }
    return CGO_OK;
}

// wait for the record of global round `id` (0-based)
int HipBackend::pipe_wait(unsigned long long id, CtlRecord &rec) {
    const int idx = (int)(id % PIPE_RING);
    if (pipe_checked_) {   // fused rounds: the record validates itself (tail_publish_record)
        static_assert(sizeof(CtlRecord) % 8 == 0, "record = 8-byte words");
        constexpr int WR = (int)(sizeof(CtlRecord) / 8);
        double words[WR];
        if (int rc = wait_checked(ctx_, ctl_seq_ + idx, id + 1, reinterpret_cast<const double *>(ctl_rec_) + (size_t)idx * WR, WR, words)) return rc;
        std::memcpy(&rec, words, sizeof(CtlRecord));
        if (rec.npts >= 0 && !ctx_->single()) {   // the round exchanged its block between the GPUs: its cost, for cgo_ctx_exchange_stats
            ctx_->xch_count++; ctx_->xch_dev_ms += (double)rec.xwait * 1e-5; ctx_->xch_dev_n++;
        }
        return CGO_OK;
    }
    if (int rc = wait_word(ctx_, ctl_seq_ + idx, id + 1)) return rc;
    rec = ((CtlRecord *)ctl_rec_)[idx];
    return CGO_OK;
}

// Before any launch that is not controller-armed: every round still in flight must be a no-op
// (the controller stops exactly where the host-side state machine leaves the fast path).
int HipBackend::pipe_drain() {
    while (pipe_done_ < pipe_enq_) {
        CtlRecord rec;
        if (int rc = pipe_wait(pipe_done_, rec)) return rc;
        pipe_done_++;
        if (rec.npts >= 0) {
            set_error("internal: the on-device controller ran a launch the host state machine did not ask for");
            return CGO_ESTATE;
        }
    }
    return CGO_OK;
}

int HipBackend::accept_dir_trial_ctl(const CtlConfig &cc, const CtlState &s0, int64_t rounds, Scal *out) {
    if (ctl_depth() <= 0) return accept_dir_trial(s0.a_acc, s0.beta, s0.a, s0.npts, out);
    if (int rc = pipe_alloc()) return rc;
    // how far to run ahead: one more round per first trial accepted in a row (host-observed)
    const int64_t ahead = std::min<int64_t>(std::min<int64_t>(ctl_depth_, pipe_streak_), rounds - 1);
    if (pipe_done_ == pipe_enq_) {  // idle: arm a new batch from the host's state
        if (ahead <= 0) { pipe_streak_++; return accept_dir_trial_keep_streak(s0, out); }
        HIPCHK(hipSetDevice(ctx_->device));
        pipe_npts_ = cc.maxp;
        k_ctl_init<<<1, 1, 0, ctx_->stream>>>((CtlDev *)ctl_dev_, cc, s0, pipe_enq_);
        HIPCHK(hipGetLastError());
        pipe_stopped_ = false;
        if (int rc = pipe_enqueue(1 + ahead)) return rc;
    }
    CtlRecord rec;
    const unsigned long long id = pipe_done_;
    if (int rc = pipe_wait(id, rec)) return rc;
    pipe_done_++;
    if (rec.npts < 0) {  // the controller had stopped before this round: the host drives it
        if (int rc = pipe_drain()) return rc;
        pipe_streak_++;
        return accept_dir_trial_keep_streak(s0, out);
    }
    if (std::memcmp(&rec.a_acc, &s0.a_acc, 8) || std::memcmp(&rec.beta, &s0.beta, 8) || rec.npts != s0.npts ||
        std::memcmp(rec.a, s0.a, 8 * (size_t)s0.npts)) {
        set_error("internal: the on-device controller and the host state machine disagree on a launch");
        return CGO_ESTATE;
    }
    const int np = pipe_npts_;
    for (int j = 0; j < s0.npts; ++j) {
        const double *q = rec.sums + RS_PER_POINT * j;
        out[j].f = q[RS_F]; out[j].gtu = q[RS_GTU]; out[j].gtgt = q[RS_GTGT]; out[j].gtg = q[RS_GTG];
        out[j].yy = q[RS_YY]; out[j].uy = q[RS_UY]; out[j].ygt = q[RS_YGT];
    }
    out[0].gu = rec.sums[RS_PER_POINT * np]; out[0].uu = rec.sums[RS_PER_POINT * np + 1];
    total_launches_++;
    pipe_served_++;
    pipe_streak_++;
    const auto &pp = pipe_prof_[(int)(id % PIPE_RING)];
    if (prof_on_) {
        prof_cnt_[KK_ACCEPT_DIR_TRIAL]++;
        prof_bytes_[KK_ACCEPT_DIR_TRIAL] = bytes_r(obj_->kind, R_ACCEPT | R_DIR | R_TRIAL, obj_->n_local, obj_->uses_param());
        if (pp.first >= 0 && pp.second == prof_gen_ && pp.first < ring_used_) {
            ring_[pp.first].kk = KK_ACCEPT_DIR_TRIAL;
            ring_[pp.first].bytes = prof_bytes_[KK_ACCEPT_DIR_TRIAL];
        }
    }
    if (!rec.accepted) pipe_stopped_ = true;
    if (!pipe_stopped_) {  // keep the device `ahead` rounds in front of the host
        // top the run-ahead up in batches (half the depth at a time) so that graph replays stay worth their launch
        const int64_t want = std::min<int64_t>(std::min<int64_t>(ctl_depth_, pipe_streak_), rounds - 1);
        const int64_t have = (int64_t)(pipe_enq_ - pipe_done_);
        if (have < want && (want - have >= (want + 1) / 2 || have == 0))
            if (int rc = pipe_enqueue(want - have)) return rc;
    }
    return CGO_OK;
}

// host-driven accept+dir+trial that does not reset the first-trial streak counter
int HipBackend::accept_dir_trial_keep_streak(const CtlState &s0, Scal *out) {
    const int64_t keep = pipe_streak_;
    const int rc = accept_dir_trial(s0.a_acc, s0.beta, s0.a, s0.npts, out);
    pipe_streak_ = keep;
    return rc;
}

void unpack_r(const double *s, int k, Scal *out, bool dir) {
    const int npts = npts_for(k);
    for (int j = 0; j < k; ++j) {
        const double *q = s + RS_PER_POINT * j;
        out[j].f = q[RS_F]; out[j].gtu = q[RS_GTU]; out[j].gtgt = q[RS_GTGT]; out[j].gtg = q[RS_GTG];
        out[j].yy = q[RS_YY]; out[j].uy = q[RS_UY]; out[j].ygt = q[RS_YGT];
    }
    if (dir) { out[0].gu = s[RS_PER_POINT * npts]; out[0].uu = s[RS_PER_POINT * npts + 1]; }
}

// ---- resident solver (cgo_resident.hpp, cgo_kernels_resident.hip.hpp) -------------------------------------------------
// Which shards: the built-in element-wise objectives under a CG β and one of the two bisection line searches, on one rank,
// while x, u (and the parameter vector) fit the LDS of the chip's CUs — one workgroup per CU at most, so that every
// workgroup of the launch is resident and their all-gather can complete.  CGO_RESIDENT=0 switches it off,
// CGO_RES_CHUNK sets the elements per workgroup (default 4096: n = 1e6 → 245 workgroups; n ≤ 4096 → ONE workgroup and no
// exchange at all), CGO_RES_POINTS the trial steps per pass (default 3).
constexpr int64_t RES_REC_CAP = 4096;     // iterations per slice at most
constexpr int64_t RES_LOG_CAP = 1 << 16;  // trial-log entries per slice

template <class Obj>
static const void *res_kernel(int npts) {
    return npts >= 7 ? (const void *)k_resident<Obj, 7> : (npts >= 3 ? (const void *)k_resident<Obj, 3> : (const void *)k_resident<Obj, 1>);
}
static const void *res_kernel_for(int obj_kind, int npts) {
    switch (obj_kind) {
    case CGO_OBJ_ROSENBROCK_CHAINED: return npts >= 3 ? (const void *)k_resident_chain<3> : (const void *)k_resident_chain<1>;   // ONE workgroup
    case CGO_OBJ_QUAD_DIAG: return res_kernel<ObjQuadDiag>(npts);
    case CGO_OBJ_ROSENBROCK_PAIRED: return res_kernel<ObjRosenPaired>(npts);
    case CGO_OBJ_BOOTH: return res_kernel<ObjBooth>(npts);
    default: return nullptr;
    }
}

int HipBackend::res_plan() {
    if (res_grid_ != 0) return res_grid_ > 0 ? res_grid_ : 0;
    res_grid_ = -1;   // decided: does not fit, unless the plan below completes
    const int64_t want = pol_.resident_chunk >= 2 ? (int64_t)(pol_.resident_chunk & ~1) : (int64_t)4096;
    const int pts = (pol_.resident_points == 1 || pol_.resident_points == 3 || pol_.resident_points == 7) ? pol_.resident_points : 3;
    res_npts_ = pts;
    const void *fn = res_kernel_for(obj_->kind, res_npts_);
    // a run-time compiled objective carries its own copy of the kernel (k_resident<UserObjective, 3>, cgo_rtc.hip)
    hipFunction_t mf = (obj_->kind == CGO_OBJ_USER && obj_->rtc) ? obj_->rtc->resident(res_npts_) : nullptr;
    if (!fn && !mf) return 0;
    const int64_t n = obj_->n_local;
    const int vecs = chain() ? 4 : (obj_->uses_param() ? 3 : 2);   // (the stencil objective: two LDS copies of x and of u)
    if (chain() && res_npts_ > 3) res_npts_ = 3;
    int max_lds = 0;
    if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx_->device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int64_t static_lds = 0;
    if (fn) {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, fn) != hipSuccess) { (void)hipGetLastError(); return 0; }
        static_lds = (int64_t)fa.sharedSizeBytes;
    } else {
        int v = 0;
        if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, mf) != hipSuccess) { (void)hipGetLastError(); return 0; }
        static_lds = v;
    }
    const int64_t avail = (int64_t)max_lds - static_lds - 512;
    int64_t chunk_max = (avail / (8 * vecs)) & ~1LL;
    if (chunk_max < 2) return 0;
    const int cus = std::min(ctx_->num_cu > 0 ? ctx_->num_cu : 256, RES_GSIZE * RES_GROUPS);   // (the two-level exchange holds 16 groups of 16)
    int64_t chunk = std::min<int64_t>(want, chunk_max);
    if (chain()) {   // the whole (padded) vector in ONE workgroup, or not at all
        chunk = n + (n & 1);
        if (chunk > chunk_max) return 0;
    }
    int64_t grid = (n + chunk - 1) / chunk;
    if (grid > cus) {   // more elements per workgroup, up to what the LDS holds
        chunk = (((n + cus - 1) / cus) + 1) & ~1LL;
        if (chunk > chunk_max) return 0;
        grid = (n + chunk - 1) / chunk;
    }
    const size_t lds = (size_t)chunk * 8 * vecs;
    if (fn && lds > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int per_cu = 0;
    if (fn) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, BLOCK, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); return 0; } }
    else if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mf, BLOCK, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); return 0; }
    if (grid > (int64_t)cus * per_cu) return 0;   // every workgroup must be resident: they wait for one another
    res_chunk_ = chunk; res_lds_ = lds; res_grid_ = (int)grid;
    return res_grid_;
}

bool HipBackend::resident_ready(const cgo_cg_config &cfg, const cgo_ls_config &ls) const {
    if (!res_on_ || !rmode_ || sys_on_ || !ctx_->single()) return false;
    if (cfg.beta.kind == CGO_BETA_LBFGS) return false;
    if (ls.kind != CGO_LS_STRONG_WOLFE_BISECTION && ls.kind != CGO_LS_WOLFE_BISECTION) return false;
    return const_cast<HipBackend *>(this)->res_plan() > 0;
}

int HipBackend::res_alloc() {
    if (res_state_) return CGO_OK;
    HIPCHK(hipSetDevice(ctx_->device));
    HIPCHK(hipHostMalloc((void **)&res_state_, sizeof(ResState), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&res_recs_, sizeof(ResRecord) * RES_REC_CAP, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&res_done_, 64, hipHostMallocDefault));
    *res_done_ = 0;
    const size_t xb = sizeof(double) * RES_XBUFS * ((size_t)res_grid_ + RES_GROUPS) * RES_WMAX;   // workgroup rows, then group rows
    HIPCHK(hipMalloc((void **)&res_xbuf_, xb));
    HIPCHK(hipMemsetD32((hipDeviceptr_t)res_xbuf_, (int)(TAIL_EMPTY & 0xFFFFFFFFull), xb / 4));
    HIPCHK(hipMalloc((void **)&res_recs_dev_, sizeof(ResRecord) * RES_REC_CAP));
    HIPCHK(hipMalloc((void **)&res_err_, 64));     // [0] error flags, [1] workgroups that have reported in
    HIPCHK(hipMemset(res_err_, 0, 64));
    if (res_grid_ > 1) {   // a multi-workgroup slice leaves x, u in these; swapped in on a good global verdict only
        if (int rc = res_xo_.alloc((size_t)obj_->n_local)) return rc;
        if (int rc = res_uo_.alloc((size_t)obj_->n_local)) return rc;
    }
    HIPCHK(hipDeviceSynchronize());
    res_round_ = 0;
    return CGO_OK;
}

int HipBackend::resident_run(const ResConfig &c, ResState &s, int64_t budget, std::vector<ResRecord> &recs, std::vector<ResLog> &log) {
    if (int rc = pipe_drain()) return rc;
    pipe_streak_ = 0;
    if (res_plan() <= 0) { set_error("internal: resident slice on a shard that does not fit"); return CGO_ESTATE; }
    if (obj_->uses_param() && !obj_->p0_set) { set_error("objective parameter vector (slot 0) was never set"); return CGO_ESTATE; }
    if (int rc = res_alloc()) return rc;
    HIPCHK(hipSetDevice(ctx_->device));
    if (c.log_on && !res_log_) {
        HIPCHK(hipHostMalloc((void **)&res_log_, sizeof(ResLog) * RES_LOG_CAP, hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&res_log_dev_, sizeof(ResLog) * RES_LOG_CAP));
    }
    ResParams P{};
    P.x = xc_; P.u = uc_; P.p0 = obj_->p0.p; P.n = obj_->n_local; P.chunk = res_chunk_; P.s0 = obj_->s0;
    const bool oop = res_grid_ > 1;
    double *xo = oop ? ((xc_ == res_xo_.p) ? res_xin_ : res_xo_.p) : xc_, *uo = oop ? ((uc_ == res_uo_.p) ? res_uin_ : res_uo_.p) : uc_;
    P.xo = xo; P.uo = uo; P.arrive = res_err_ + 1;
    P.inject = -1;
    if (res_slices_ == 0) { if (const char *e = getenv("CGO_RES_INJECT_GIVEUP")) P.inject = atoi(e); }   // test hook: first slice only
    P.cfg = c; P.cfg.npts = res_npts_;
    P.st = s;
    if (P.st.ncache > res_npts_) P.st.ncache = res_npts_;   // (a wider host launch left more trial results than a pass of this width keeps)
    P.budget = std::min<int64_t>(budget, RES_REC_CAP);
    P.st_out = res_state_; P.recs = res_recs_dev_; P.log = res_log_dev_; P.log_cap = c.log_on ? RES_LOG_CAP : 0;
    P.recs_host = res_recs_; P.log_host = res_log_;
    P.xbuf = res_xbuf_; P.round0 = res_round_; P.err = res_err_;
    P.done_seq = res_done_; P.seq = ++res_seq_;
    static const bool timing = getenv("CGO_RES_TIMING") != nullptr;
    P.timing = timing ? 1 : 0;
    const void *fn = res_kernel_for(obj_->kind, res_npts_);
    void *args[] = {&P};
    const double h0 = timing ? now_ns() : 0.0;
    if (int rc = prof_begin(KK_RESIDENT)) return rc;
    if (fn) HIPCHK(hipLaunchKernel(fn, dim3(res_grid_), dim3(BLOCK), args, res_lds_, ctx_->stream));
    else HIPCHK(hipModuleLaunchKernel(obj_->rtc->resident(res_npts_), res_grid_, 1, 1, BLOCK, 1, 1, (unsigned)res_lds_, ctx_->stream, args, nullptr));
    if (int rc = prof_end()) return rc;
    total_launches_++;
    const double h1 = timing ? now_ns() : 0.0;
    if (int rc = wait_word(ctx_, res_done_, res_seq_)) return rc;
    const double h2 = timing ? now_ns() : 0.0;
    if (timing) fprintf(stderr, "[cgo resident] host: enqueue %.1f us, wait for the slice %.1f us\n", (h1 - h0) * 1e-3, (h2 - h1) * 1e-3);
    s = *res_state_;
    res_round_ += (unsigned long long)s.passes;
    res_slices_++;
    {
        if (timing) fprintf(stderr, "[cgo resident] slice: %lld iterations, %lld passes, grid %d x %lld elements, reason %d: %.1f us in all; per pass compute %.2f, "
                                 "workgroup reduce %.2f, exchange %.2f us; outside the passes %.2f us per iteration (machine %.2f, evals incl. passes %.2f, post %.2f); shader clock %.0f MHz\n",
                         (long long)s.done, (long long)s.passes, res_grid_, (long long)res_chunk_, (int)s.reason, s.t_total * 1e-2,
                         s.t_compute * 1e-2 / std::max<double>(s.passes, 1), s.t_reduce * 1e-2 / std::max<double>(s.passes, 1),
                         s.t_exchange * 1e-2 / std::max<double>(s.passes, 1),
                         (s.t_total - s.t_compute - s.t_reduce - s.t_exchange) * 1e-2 / std::max<double>(s.done, 1),
                         s.t_machine * 1e-2 / std::max<double>(s.done, 1), s.t_eval * 1e-2 / std::max<double>(s.done, 1), s.t_post * 1e-2 / std::max<double>(s.done, 1),
                         (double)s.t_cycles / std::max<double>((double)s.t_total * 1e-2, 1e-9));
    }
    if (s.reason == RES_ERROR) {
        // The exchange gave up: some workgroup of the launch was not running while the others waited for its row.  That
        // happens when ANOTHER process's kernels hold CUs (two persistent launches can each be partially resident and wait
        // for workgroups the other one's keep out).  Nothing is lost: a slice writes x, u back only when it ends well, so
        // the state is still that of the slice's start — hand the whole slice to the launch-per-trial engine and keep this
        // solver off the resident path from here on (correct under any sharing of the GPU, at the old speed).
        HIPCHK(hipStreamSynchronize(ctx_->stream));
        const size_t xb = sizeof(double) * RES_XBUFS * ((size_t)res_grid_ + RES_GROUPS) * RES_WMAX;
        HIPCHK(hipMemsetD32((hipDeviceptr_t)res_xbuf_, (int)(TAIL_EMPTY & 0xFFFFFFFFull), xb / 4));
        HIPCHK(hipMemset(res_err_, 0, 64));
        res_round_ = 0;
        res_on_ = false;
        res_gave_up_++;
        s = P.st;   // the state the slice started from
        s.done = 0; s.log_len = 0; s.evals = 0; s.passes = 0; s.reason = RES_HOST;
        recs.clear(); log.clear();
        return CGO_OK;
    }
    if (oop && s.done > 0) {   // a good slice, by the verdict of ALL its workgroups: its x, u become the iterate
        res_xin_ = xc_; res_uin_ = uc_;
        xc_ = xo; uc_ = uo;
    }
    res_iters_ += s.done;
    recs.assign(res_recs_, res_recs_ + s.done);
    if (c.log_on) log.assign(res_log_, res_log_ + s.log_len); else log.clear();
    // state moved once per slice: load x, u (+ D) and store x, u
    if (prof_on_) prof_commit(KK_RESIDENT, 8.0 * (double)obj_->n_local * (double)((obj_->uses_param() ? 3 : 2) + (s.done > 0 ? 2 : 0)));
    return CGO_OK;
}

}  // namespace cgo

#ifdef CGO_STAMPS
// diagnostic build: the per-workgroup stamps of the last k_cg launch (cgo_kernels_cg.hip.hpp); the caller has synchronised
extern "C" int cgo_debug_stamps(unsigned long long *out, int words) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cgo::dev::cgo_stamps), (size_t)words * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
