// cgo_hip_backend.hpp — the product's only VecBackend: device-resident solver
// state in HBM + the fused gfx950 launches of cgo_kernels.hip.hpp.
#pragma once

#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "cgo_engine.hpp"
#include "cgo_rtc.hpp"

namespace cgo {

namespace dev { struct CtlArgs; struct Tail; }

void set_error(const std::string &msg);
const char *get_error();

// RAII device allocation
struct DevBuf {
    double *p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t count);
    void release();
};

struct HipCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::unique_ptr<Comm> comm;  // null = single rank
    // reduction scratch shared by every launch on this ctx (one solve at a time)
    double *partials = nullptr;      // [MAX_GRID][NR]
    double *partials2 = nullptr;     // [MAX_GRID/64][NR]  second-stage rows
    double *out_dev = nullptr;       // [NS] local sums
    unsigned int *tickets = nullptr; // [64 + 1] arrival counters of the fused reduction tail (zero between launches) + 1 error counter
    double *partials_f = nullptr;    // [MAX_GRID][NR7] rows of fused launches: every slot holds TAIL_EMPTY between launches
    double *partials2_f = nullptr;   // [64][NR7] their group rows, likewise
    bool pub_checked = false;        // the last publisher was a fused launch: its host block validates itself (tail_check_term)
    bool tail_strict = false;        // CGO_TAIL_STRICT=1: formal system-scope fence for the host block (finish_tail)
    bool fused_tail = true;          // k_cg / k_chain launches finish their own sums (CGO_FUSED_TAIL=0: finalize launches)
    double *gather_dev = nullptr;    // [world][NS]
    double *host_pinned = nullptr;   // [max(world,1)][NS]
    unsigned long long *host_seq = nullptr;  // pinned; k_finalize publishes the launch sequence number here
    unsigned long long seq = 0;      // sequence number of the last launch that produced sums
    bool host_publish = true;        // poll pinned memory instead of D2H copy + stream sync
    // exchange diagnostics (cgo_ctx_exchange_stats)
    long long xch_count = 0;         // launches whose sums crossed ranks
    double xch_peer_wait_ns = 0;     // mailbox: own block visible → last peer's block visible
    hipEvent_t xev0 = nullptr, xev1 = nullptr;  // device-side duration of all-gather + publish, every 4th exchange
    bool xev_pending = false;
    double xch_dev_ms = 0; long long xch_dev_n = 0;
    void xch_collect();              // fold a pending event pair into xch_dev_ms
    bool force_gather = false;       // debug (CGO_FORCE_GATHER=1): run the multi-rank exchange path even with one rank
    unsigned long long solver_epoch = 0;   // solvers created on this context so far (the same on every rank: replicated control flow)
    bool dev_exchange() const { return comm && world() > 1 && comm->dev_mailbox(0) != nullptr; }   // device mailboxes connected
    bool single() const { return world() == 1 && !force_gather; }
    bool shm() const { return comm && comm->shm_slot_host(0, 0) != nullptr; }
    // where the finalize kernel of launch `seq` publishes (nullptr = no host publish for this launch)
    void pub_target(double **out, unsigned long long **seqw);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // the (x, u) buffers the last solver's placement search kept, parked here when that solver goes so that the next solver
    // of the same size (a rerun chain, the next centering step) takes them over instead of searching again
    DevBuf placed_x, placed_u;
    int64_t placed_n = 0;
    double placed_first_us = 0.0, placed_best_us = 0.0;
    int placed_candidates = 0;
    std::string arch;
    int num_cu = 0;
    ~HipCtx();
    int init(int device);
    int ensure_gather();
    int rank() const { return comm ? comm->rank : 0; }
    int world() const { return comm ? comm->world : 1; }
};

struct HipObjective {
    HipCtx *ctx = nullptr;
    int kind = 0;
    int64_t n_global = 0, offset = 0, n_local = 0;
    DevBuf p0;
    bool p0_set = false;
    double s0 = 0.0;
    std::shared_ptr<RtcModule> rtc;  // CGO_OBJ_USER: the run-time compiled kernels
    bool user_has_param = false;
    bool user_cheap = false;  // cgo_objective_set_cost_class: seven trial steps per launch
    int users = 0;            // live solvers on this objective (the placement search moves the parameter vector only for a sole user)
    // CGO_OBJ_HOST: the reference's closure contract f = fdf!(g, x) on host vectors (cgo_objective_create_callback)
    cgo_fdf_fn host_fn = nullptr;
    void *host_user = nullptr;
    double *host_x = nullptr, *host_g = nullptr;   // pinned staging, n_local doubles each
    bool host_closure() const { return kind == CGO_OBJ_HOST; }
    ~HipObjective();
    bool uses_param() const { return kind == CGO_OBJ_QUAD_DIAG || (kind == CGO_OBJ_USER && user_has_param); }
    bool two_phase() const { return kind == CGO_OBJ_LSE; }
};

class HipBackend : public VecBackend {
  public:
    HipBackend(HipCtx *ctx, HipObjective *obj);
    ~HipBackend() override;
    int alloc();
    int place();   // placement search for HBM-bound sizes (after the launch policy is set)
    int64_t n_local() const override { return obj_->n_local; }
    int set_x0_host(const double *x0) override;
    int set_x0_device(const double *x0_dev);
    int download_device(double *x_dev, double *g_dev);
    int tail_errors();
    int set_x0_fill(int kind, uint64_t seed, double lo, double hi) override;
    int init_eval(Scal &out) override;
    // 3-point launches pay a 24-slot reduction: worth it once a saved launch is worth more than
    // that (measured: n = 1e6 loses 20 %, n = 1e7 gains 70 %)
    // Trial steps per launch (policy set by the C API from the objective's cost class; measured in
    // scripts/ab_points.sh, ab_small.sh, ab_small2.sh).  The on-device controller understands 3-point rows.
    int max_points() const override {
        if (!rmode_ || obj_->n_local < multi_min_n_) return 1;
        return policy_points();
    }
    int policy_points() const {
        if (!rmode_ || obj_->n_local < multi_min_n_) return 1;
        if (obj_->n_local >= band3_lo_ && obj_->n_local < band3_hi_) return 3;
        if (obj_->n_local >= multi7_min_n_) return 7;
        return obj_->n_local >= multi5_min_n_ ? 5 : 3;
    }
    int trial(const double *a, int k, Scal *out) override;
    int accept_dir_trial(double a_acc, double beta, const double *a, int k, Scal *out) override;
    int ctl_depth() const override;
    int accept_dir_trial_ctl(const CtlConfig &cc, const CtlState &s0, int64_t rounds, Scal *out) override;
    void set_ctl_depth(int d) { ctl_depth_ = d < 0 ? 0 : (d > 32 ? 32 : d); }
    void set_policy(const cgo_solver_policy &p);          // the resolved policy (before alloc())
    const cgo_solver_policy &policy() const { return pol_; }
    double big_bytes(bool read_only = false) const;       // pure-HBM streaming threshold of this solver
    int prepare_controller();   // its device / pinned blocks and the first launch of its kernels, outside the first armed iteration
    int64_t ctl_served() const { return pipe_served_; }
    int64_t ctl_graph_rounds() const { return graph_rounds_; }
    void placement_info(double *first_us, double *best_us, int *candidates) const { *first_us = place_first_us_; *best_us = place_best_us_; *candidates = place_candidates_; }
    double placement_cap_bytes() const { return place_cap_bytes_; }
    void set_ctl_graph(bool on) { graph_on_ = on; }
    // resident solver (cgo_kernels_resident.hip.hpp): whole iterations in one launch while the shard fits the LDS of the chip
    bool resident_ready(const cgo_cg_config &cfg, const cgo_ls_config &ls) const override;
    int resident_run(const ResConfig &c, ResState &s, int64_t budget, std::vector<ResRecord> &recs, std::vector<ResLog> &log) override;
    void set_resident(bool on) { res_on_ = on; }
    bool resident_enabled() const { return res_on_; }
    int ctl_depth_setting() const { return ctl_depth_; }
    int64_t resident_iters() const { return res_iters_; }
    int64_t resident_slices() const { return res_slices_; }
    int64_t resident_gave_up() const { return res_gave_up_; }
    int64_t push_count(int kind) const { return push_counts_[kind]; }   // 0 speculated, 1 fused (g⁺ formed in the push), 2 plain
    bool sys_supported() const override { return rmode_; }
    int sys_begin() override;
    int sys_project(double a, double m, Scal &out) override;
    int sys_commit() override;
    int dir_trial(double beta, const double *a, int k, Scal *out) override;
    int accept_dir(double a_acc, double beta, Scal &out) override;
    int accept_only(double a_acc) override;
    int reset_dir(Scal &out) override;
    int upg_sumsq(double &out) override;
    int lbfgs_alloc(int slots) override;
    int lbfgs_gram_max_pairs() const override { return gram_on_ ? 12 : 0; }
    int lbfgs_push_gram(double a_x, double a_s, int slot, const int *prev, int count, GramOut &out) override;
    bool lbfgs_push_materializes(double a_x) override;
    bool lbfgs_push_spec(double a_x, double a_s, int slot, const int *prev, int count, GramOut &out) override;
    int lbfgs_push_commit(bool direction_follows) override;
    int lbfgs_direction_gram(const int *slots, const double *cy, const double *cs, int count, double cg,
                             Scal &out) override;
    bool lbfgs_direction_gram_can_fuse_trial() const override;
    int lbfgs_direction_gram_trial(const int *slots, const double *cy, const double *cs, int count, double cg, double a_trial,
                                   Scal &dir, Scal &trial) override;
    int lbfgs_push(double a_x, double a_s, int slot, double &sy, double &yy) override;
    int lbfgs_direction(const int *slots, const double *rho, int count, double gamma,
                        Scal &out) override;
    bool two_phase() const override { return obj_->two_phase(); }
    int materialize(Scal &out) override;
    void discard_pending() override { push_pending_ = false; push_lite_pending_ = false; lite_deferred_ = false; spec_valid_ = false; spec_unmat_ = false; }
    int download(double *x, double *g) override;
    int scaled_norm_parts(int which, double a_trial, double &maxabs, double &scaled_ss, bool &has_nan) override;
    void profile_enable(bool on) override;
    void profile_reset() override;
    void profile_get(int kind, int64_t *launches, double *ms, double *bytes) override;
    int64_t launches() const override { return total_launches_; }
    void set_need_beta(bool b) { need_beta_ = b; }
    // gradient-free multi-point CG kernels (cgo_kernels_cg.hip.hpp): element-wise objective + CG β
    void set_rmode(bool on) { rmode_ = on; }
    void set_multi_min_n(int64_t n) { multi_min_n_ = n; }
    void set_multi5_min_n(int64_t n) { multi5_min_n_ = n; }
    void set_multi7_min_n(int64_t n) { multi7_min_n_ = n; }
    void set_three_point_band(int64_t lo, int64_t hi) { band3_lo_ = lo; band3_hi_ = hi; }
    bool rmode() const { return rmode_; }
    std::string kernel_symbol(int kernel_kind) const;

    // raw single-launch helpers used by the kernel-level C entry points
    static int run_dir(HipCtx *ctx, double *u_host, const double *g_host, double beta, int64_t n,
                       double *out2);
    static int run_beta_partials(HipCtx *ctx, const double *gn, const double *g, const double *u,
                                 int64_t n, double *out9);
    static int run_trial(HipObjective *obj, const double *x, const double *u, double a,
                         double *gn_out, double *out2);
    static int run_eval(HipObjective *obj, const double *x, double *g_out, double *f);
    static int bench_kernel(HipCtx *ctx, HipObjective *obj, int kernel_kind, int64_t n, int reps,
                            double *ms, double *bytes);
    static int bench_stream_mix(HipCtx *ctx, int64_t n, int reps, double *median_us, double *best_us);

  private:
    int launch(int kk, int mode, double a_acc, double beta, double a_trial, bool fetch,
               double *sums /*[NS] global*/);
    HipCtx *ctx_;
    HipObjective *obj_;
    DevBuf x_, u_, ga_, gb_;
    // Ping-pong iterate/direction buffers of the gradient-free family's pure-HBM (BIG) launches: such a launch
    // writes the updated x / u into the OTHER buffer of the pair and the pointers swap — reading and writing the same
    // arrays in one stream costs ≈ 10 % of HBM bandwidth.  Allocated on first use, only if they fit.
    DevBuf x2_, u2_;
    double *uc_ = nullptr;                     // current direction (u_.p or u2_.p)
    double *xalt_ = nullptr, *ualt_ = nullptr; // the other buffer of each pair
    int pingpong_ = -1;                        // −1: not decided yet, 0: in place, 1: ping-pong
    bool sys_on_ = false;                      // solvesystem owns a second iterate buffer of its own: in place
    // chained Rosenbrock (stencil objective, cgo_kernels_chain.hip.hpp): x / u always advance out of place, and the
    // two elements beyond each shard boundary travel in the scalar block (slots 10–17 of every rank's row)
    bool chain() const { return obj_->kind == CGO_OBJ_ROSENBROCK_CHAINED; }
    double halo_xl_[2] = {0, 0}, halo_ul_[2] = {0, 0}, halo_xr_[2] = {0, 0}, halo_ur_[2] = {0, 0};
    int launch_chain_kernel(int mode, double a_acc, double beta, const double *a, int k, int npts, bool big, int grid, const dev::Tail &tail);
    bool tail_fused(int grid) const;
    bool pipe_fused(int grid) const;
    dev::Tail make_tail(bool on);
    bool pingpong_ready();
    // placement search for pure-HBM problem sizes: which physical buffers x, u (and D) live in
    int tune_placement();
    double place_first_us_ = 0.0, place_best_us_ = 0.0;   // the mix on the buffers as allocated / on the chosen ones
    int place_candidates_ = 0;
    double place_cap_bytes_ = 0.0;          // transient memory the placement search may use (policy.placement_max_bytes or a quarter of the free memory)
    bool placed_ = false;
    int ensure_ga();   // gradient buffer A on first use
    int ensure_gb();   // gradient buffer B / solvesystem's second iterate on first use
    double *xc_ = nullptr, *xn_ = nullptr;  // current iterate / solvesystem's x_next (swapped by sys_commit)
    double *g_ = nullptr, *gt_ = nullptr;  // rotate between ga_/gb_ (kills optim.jl:139's copy)
    bool need_beta_ = true;
    bool rmode_ = false;
    int64_t multi_min_n_ = 3000000;
    int64_t multi5_min_n_ = INT64_MAX;
    int64_t multi7_min_n_ = INT64_MAX;
    int64_t band3_lo_ = 0, band3_hi_ = 0;  // sizes inside [lo, hi) stay at three points
    int launch_r(int kk, int mode, double a_acc, double beta, const double *a, int k, bool fetch, double *sums);
    int launch_r_kernel(int kk, int mode, double a_acc, double beta, const double *a, int k, int npts,
                        const struct dev::CtlArgs *ctl, int *grid_out);
    // on-device controller (cgo_ctl.hpp): rounds armed on the device ahead of the host
    int ctl_depth_ = 0;  // set by the C API per objective class / CGO_CTL_DEPTH
    unsigned long long epoch_ = 0;          // this solver's number on its context (device-mailbox block numbers carry it)
    void *ctl_dev_ = nullptr;               // CtlDev in HBM
    void *ctl_rec_ = nullptr;               // CtlRecord[PIPE_RING], pinned host
    unsigned long long *ctl_seq_ = nullptr; // [PIPE_RING], pinned host
    unsigned long long pipe_enq_ = 0, pipe_done_ = 0;  // rounds enqueued / consumed (global counters)
    bool pipe_stopped_ = false;
    bool pipe_checked_ = false;             // rounds in flight publish self-validating records (fused rounds, not CGO_TAIL_STRICT)
    bool ctl_fused_ = true;                 // policy.controller_fused = 0: armed rounds keep their reduce + controller launches
    cgo_solver_policy pol_;                 // resolved policy of this solver (set_policy)
    int pipe_npts_ = 1;                     // kernel variant (1, 3, 5, 7 trial points) of the rounds in flight
    int64_t pipe_streak_ = 0;               // accept+dir+trial launches in a row = first trials accepted in a row
    int64_t pipe_served_ = 0;
    std::vector<std::pair<int, unsigned>> pipe_prof_;  // profiling slot (index, generation) of each round
    unsigned prof_gen_ = 0;
    int pipe_alloc();
    int pipe_round_kernels();
    int pipe_enqueue_round();
    int pipe_launch_graph(int rounds);
    int pipe_enqueue(int64_t count);
    struct PipeGraph { void *exec; int rounds, npts; double *x, *u; const double *p0; int64_t n; };
    std::vector<PipeGraph> graphs_;      // instantiated hipGraphs of 2 / 4 / 8 controller rounds
    bool graph_on_ = false;              // CGO_CTL_GRAPH=1: batches of armed rounds replay from instantiated hipGraphs (measured 3–8 % slower than kernel-by-kernel enqueue)
    bool capturing_ = false;
    unsigned long long pipe_batches_ = 0;
    int64_t graph_rounds_ = 0;
    int pipe_wait(unsigned long long id, CtlRecord &rec);
    int pipe_drain();
    int accept_dir_trial_keep_streak(const CtlState &s0, Scal *out);
    // resident solver state
    bool res_on_ = true;
    int res_grid_ = 0, res_npts_ = 3;
    int64_t res_chunk_ = 0;
    size_t res_lds_ = 0;
    int res_plan();                          // grid, chunk, LDS bytes for this shard; 0 workgroups = does not fit
    int res_alloc();
    ResState *res_state_ = nullptr;          // pinned
    ResRecord *res_recs_ = nullptr;          // pinned [RES_REC_CAP]
    ResLog *res_log_ = nullptr;              // pinned [RES_LOG_CAP], allocated when a log is first asked for
    ResRecord *res_recs_dev_ = nullptr;      // device twins: the kernel's leader writes here, workgroup 0 copies out at the end
    ResLog *res_log_dev_ = nullptr;
    double *res_xbuf_ = nullptr;             // device
    unsigned int *res_err_ = nullptr;        // device: [0] error flags of the exchange, [1] workgroups that have reported in
    DevBuf res_xo_, res_uo_;                 // where a multi-workgroup slice leaves x, u (swapped in on a good GLOBAL verdict only)
    double *res_xin_ = nullptr, *res_uin_ = nullptr;   // the pair those were swapped against (the next slice's output)
    unsigned long long *res_done_ = nullptr; // pinned
    unsigned long long res_seq_ = 0, res_round_ = 0;
    int64_t res_iters_ = 0, res_slices_ = 0, res_gave_up_ = 0;
    bool prof_on_ = false;
    struct ProfSlot { hipEvent_t e0 = nullptr, e1 = nullptr; int kk = -1; double bytes = 0; };
    std::vector<ProfSlot> ring_;
    int ring_used_ = 0;
    int prof_slot(hipEvent_t *e0, hipEvent_t *e1);
    int prof_begin(int kk);
    int prof_end();
    void prof_commit(int kk, double bytes);
    void prof_flush();
    bool prof_pick(int kk);             // is the launch about to be issued one of the timed ones?
    int prof_every_ = 4;                // below n_local = 3e7: time every 4th launch, count all
    int64_t prof_tick_[KK_COUNT] = {};
    bool prof_cur_ = false;
    int64_t prof_cnt_[KK_COUNT] = {};   // launches (all)
    int64_t prof_n_[KK_COUNT] = {};     // launches timed
    double prof_ms_[KK_COUNT] = {};
    double prof_bytes_[KK_COUNT] = {};
    int64_t total_launches_ = 0;
    // L-BFGS ring in HBM
    int qn_m_ = 0;
    bool gram_on_ = true;   // CGO_LBFGS_TWO_LOOP=1: chained two-loop launches instead of the Gram form
    DevBuf qn_S_, qn_Y_;
    double *qn_alpha_dev_ = nullptr;
    double qn_sgt_ = 0.0;   // Σ s·g⁺ of the last push (global)
    int qn_sgt_slot_ = -1;
    bool push_pending_ = false, fuse_grad_ = true;   // fused push of the log-sum-exp objective: x', g⁺ written, pointers not swapped yet
    double *push_xo_ = nullptr;
    // one ring pass per iteration: the sums the direction pass took at its speculated first trial (lbfgs_direction_spec)
    bool spec_on_ = false, spec_valid_ = false, push_lite_pending_ = false;
    bool spec_fuse_push_ = true, lite_deferred_ = false, spec_unmat_ = false;   // the state update of an accepted speculated trial left to the next direction pass
    int flush_lite();
    double spec_s_[64] = {}, spec_Mr_ = 0.0, spec_Sr_ = 1.0, spec_a_ = 0.0, spec_dphi_ = 0.0;
    int64_t spec_refreshed_ = 0;   // speculated trials whose statistics were taken again with the true maximum
    int spec_count_ = 0, spec_slots_[16] = {};
    double lite_a_ = 0.0, lite_as_ = 0.0, lite_M_ = 0.0, lite_S_ = 1.0;
    int lite_slot_ = 0;
    int64_t push_counts_[3] = {0, 0, 0};
    int lbfgs_direction_spec(const int *slots, const double *cy, const double *cs, int count, double cg, double a_trial, Scal &dir, Scal &trial);
    int lbfgs_push_lite();
    int chain_sums(int grid, int slot, const double **dot_ptr, int *dot_count, double *dot_host);
    // two-phase (LSE) state of the most recent trial
    double lse_a_ = 0.0, lse_M_ = 0.0, lse_S_ = 1.0;
    bool lse_have_ = false;   // (lse_M_, lse_S_) are the statistics of a point on the current line: usable as a fixed reference
    int lse_stats(int mode, double a_acc, double beta, double a_trial, Scal &out, bool dir);
    int lse_grad(bool init, double a, Scal &out);
    // host-closure objective: xp → pinned host, fdf!, g⁺ → device, then the getβ sums kernel (f rides in its S_F slot)
    int host_trial(double a, bool init, Scal &out);
};

// low-level launcher shared by the backend and the raw helpers
int launch_fused(HipCtx *ctx, int obj_kind, int mode, const void *kparams, int64_t n,
                 bool timed = false, const HipObjective *obj = nullptr, hipEvent_t e0 = nullptr,
                 hipEvent_t e1 = nullptr, double big_forced = -1.0 /* < 0: no solver — the CGO_BIG_BYTES experiment override, else the library's thresholds */);
int grid_for(int64_t n);
double bytes_for(int obj_kind, int mode, int64_t n, bool has_param = false);
enum MergeKind { MERGE_SUM = 0, MERGE_LSE = 1, MERGE_MAX0 = 2 };
int fetch_sums(HipCtx *ctx, double *sums, int merge = MERGE_SUM, int ns = 10, double *raw = nullptr);  // raw: every rank's block, [world][ns]
int finalize_launch(HipCtx *ctx, int grid, bool lse);
int finalize_rows(HipCtx *ctx, int rows, int ns, bool canon = false);
int fill_device(HipCtx *ctx, double *v, int64_t n, int64_t offset, int kind, uint64_t seed, double lo,
                double hi);

Comm *make_rccl_comm(HipCtx *ctx, int rank, int world, const void *unique_id128);
Comm *make_callback_comm(int rank, int world, cgo_allgather_fn fn, void *user);
Comm *make_shm_comm(HipCtx *ctx, int rank, int world, const char *name, int create);
int rccl_unique_id(void *out128);
bool rccl_available();

}  // namespace cgo
