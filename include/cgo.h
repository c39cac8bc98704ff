/*
 * cgo.h — C ABI of the MI355X-native inner-iteration engine for
 * ConjugateGradientOptim.jl's nonlinear-CG / quasi-Newton hot path.
 *
 * This is the drop-in boundary: a Julia host (`ccall`), or any other FFI,
 * binds exactly these symbols.  Plain pointers and sizes only; no C++ or torch
 * types; no exceptions cross it.  Every entry point names the reference
 * interface it replaces (paths relative to the reference repository).
 *
 * Conventions
 *   - return value: 0 = CGO_OK, otherwise an API error (bad argument, failed
 *     config assertion, HIP/RCCL error); cgo_last_error() has the message.
 *     The reference throws only on config @asserts; numerical outcomes are
 *     status symbols.  Same here: the numerical outcome is cgo_results.status.
 *   - all vectors are Float64, contiguous.  "local" sizes refer to this
 *     rank's contiguous shard [offset, offset + n_local) of the n_global state.
 *   - host buffers are caller-owned; device memory is owned by the ctx.
 *   - one solve per ctx at a time; distinct ctx may be used concurrently.
 *   - there is NO CPU fallback: without a usable HIP device cgo_ctx_create
 *     fails with CGO_ENODEV.
 */
#ifndef CGO_H
#define CGO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGO_VERSION 100

/* ---- API error codes --------------------------------------------------- */
enum {
    CGO_OK = 0,
    CGO_EINVAL = 1,  /* bad argument, or a reference @assert would have fired */
    CGO_EHIP = 2,    /* HIP runtime error */
    CGO_ECOMM = 3,   /* RCCL / communicator error */
    CGO_ENODEV = 4,  /* no usable gfx950 device */
    CGO_ESTATE = 5,  /* call sequence error (e.g. iterate before start) */
    CGO_ENOMEM = 6
};

/* ---- numerical outcome: one integer per reference status Symbol -------- */
enum {
    CGO_INCOMPLETE = 0,                                /* src/engine/optim.jl:39  */
    CGO_SUCCESS = 1,                                   /* src/engine/optim.jl:64  */
    CGO_INCREASING_OBJECTIVE = 2,                      /* src/engine/optim.jl:76  */
    CGO_NON_FINITE_OBJECTIVE_OR_GRADIENT_PROPOSED = 3, /* src/engine/optim.jl:118 */
    CGO_MAX_ITERS_REACHED = 4,                         /* src/engine/optim.jl:168 */
    CGO_NON_DESCENT_SEARCH_DIRECTION = 5,              /* src/linesearch/nocedal.jl:62, wolfe.jl:42 */
    CGO_LINESEARCH_A_MAX_OVERFLOW = 6,                 /* src/linesearch/nocedal.jl:148 */
    CGO_LINESEARCH_MAX_ITERS_REACHED = 7,              /* nocedal.jl:157, wolfe.jl:164 */
    CGO_ZOOM_MAX_ITERS_REACHED = 8,                    /* nocedal.jl:208 */
    CGO_ACCEPTED_NON_FINITE_ITERATE = 9,               /* wolfe.jl:37  */
    CGO_CANNOT_FIND_INITIAL_FEASIBLE_STEP = 10,        /* wolfe.jl:64  */
    CGO_MAX_STEP_LENGTH_REACHED = 11,                  /* wolfe.jl:111 */
    CGO_CANNOT_FIND_FEASIBLE_STEP = 12,                /* wolfe.jl:157 */
    CGO_STEP_BRACKET_PRECISION_ISSUE = 13,             /* wolfe.jl:131 (unreachable in the reference) */
    CGO_BISECTION_LOWER_BOUND_LARGER_THAN_PROPOSED_STEP = 14, /* wolfe.jl:187 */
    CGO_FEASIBLE = 15,                                 /* wolfe.jl:197 */
    CGO_INFEASIBLE = 16,                               /* wolfe.jl:206 */
    CGO_NON_FINITE_STEP_PROPOSED = 17,                 /* geometric.jl:129 */
    CGO_PROPOSED_STEP_SAME_AS_CURRENT_STEP = 18,       /* geometric.jl:133 */
    CGO_LINESEARCH_FAILED = 19,                        /* solve_system.jl:139 — where the reference itself
                                                        * throws UndefVarError (solve_system.jl:55) */
    CGO_NUM_STATUS = 20
};

/* ---- βConfig subtypes (src/types.jl:5-7; src/cg_flavours.jl) ----------- */
enum {
    CGO_BETA_HAGER_ZHANG = 0,      /* HagerZhang            cg_flavours.jl:83-108  */
    CGO_BETA_YUAN_WANG_SHENG = 1,  /* YuanWangSheng{T}(μ)   cg_flavours.jl:46-79   */
    CGO_BETA_SALLEH_ALHAWARAT = 2, /* SallehAlhawarat       cg_flavours.jl:130-151 */
    CGO_BETA_LIU_STORREY = 3,      /* LiuStorrey            cg_flavours.jl:154-170 */
    CGO_BETA_POLAK_RIBIERE = 4,    /* new CGβConfig (stub at cg_flavours.jl:173-174) */
    CGO_BETA_HESTENES_STIEFEL = 5, /* new CGβConfig (commented at cg_flavours.jl:110-127) */
    CGO_BETA_DAI_YUAN = 6,         /* new CGβConfig */
    CGO_BETA_LBFGS = 7,            /* new QNβConfig; dispatch contract src/qn_flavours.jl:5-48 */
    CGO_BETA_BROYDEN_FAMILY = 8    /* BroydenFamily{T}(θ, B) — src/qn_flavours.jl:53-90.  Its update solves
                                    * s = B\y first (:81), hence Bs = y, sBs = s·y, v = 0 and B_new = B up to
                                    * rounding (:83-87): B stays the identity it is initialised/reset to (:13-15,
                                    * :33-36) and u = B\(−g) is steepest descent for every θ.  The engine takes
                                    * u = −g exactly, without the dense n×n matrix and its O(n³) solves; the oracle
                                    * carries the dense algebra and agrees to ≈ 1e-15. */
};

/* ---- LineSearchConfig subtypes (src/types.jl:1) ------------------------ */
enum {
    CGO_LS_STRONG_WOLFE_BISECTION = 0, /* StrongWolfeBisection{T}  nocedal.jl:3-11 */
    CGO_LS_WOLFE_BISECTION = 1,        /* WolfeBisection{T,CT}     wolfe.jl:6-11   */
    CGO_LS_BACKTRACKING = 2            /* Backtracking{T,CT}       geometric.jl:15-20 (restated bug for bug) */
};
enum {
    CGO_COND_WOLFE = 0,        /* Wolfe{T}(c1,c2)              wolfe.jl:259-262 */
    CGO_COND_YUAN_WEI_LU = 1,  /* YuanWeiLuWolfe{T}(c1,c2,δ1)  wolfe.jl:213-217 */
    CGO_COND_ARMIJO = 2        /* Armijo{T}(c1)                geometric.jl:159-162 */
};

/* ---- device objective descriptors (replace the `fdf!` closure of
 *      src/engine/optim.jl:25 and src/cg_utils.jl:19 on the GPU path) ----- */
enum {
    CGO_OBJ_QUAD_DIAG = 0,         /* f = ½ Σ D_i x_i²; param vector slot 0 = D     */
    CGO_OBJ_ROSENBROCK_PAIRED = 1, /* f = Σ_j 100(x_{2j}−x_{2j−1}²)² + (1−x_{2j−1})² */
    CGO_OBJ_BOOTH = 2,             /* examples/helpers/test_funcs.jl:3-12 (n = 2)   */
    CGO_OBJ_LSE = 3,               /* f = log Σ e^{x_i} + ½λ‖x‖²; scalar slot 0 = λ  */
    CGO_OBJ_USER = 4,              /* user-supplied element-wise source, see cgo_objective_create_from_source */
    CGO_OBJ_HOST = 5,              /* a host closure f = fdf!(g, x), see cgo_objective_create_callback */
    CGO_OBJ_ROSENBROCK_CHAINED = 6 /* f = Σ_{i<N−1} (1−x_i)² + 100(x_{i+1}−x_i²)² — examples/helpers/test_funcs.jl:50-57 (value;
                                    * BASELINE config 1's second form).  A 3-point STENCIL objective: neighbours are
                                    * re-read from cache, x/u advance out of place, and shard boundaries carry a 2-element
                                    * halo inside the per-launch scalar block.  Any N ≥ 2.
                                    * LAUNCH POLICY, fenced (round 4): this objective runs host-driven launches of ONE or THREE
                                    * trial points — cgo_solver_policy.points > 3 is clamped to 3, controller_depth is ignored (no
                                    * on-device controller), CG β kinds only; the resident solver takes it in ONE workgroup, i.e.
                                    * up to ≈ 4 800 elements on a single rank (x, u in two LDS copies each) — larger or sharded
                                    * stencil problems keep a launch per line-search step.  5/7-point stencil launches and a
                                    * multi-workgroup resident form (halos between workgroups) are NOT built. */
};

/* initial-iterate fills done on the device (global index aware) */
enum {
    CGO_FILL_CONSTANT = 0,   /* v_i = lo                                  */
    CGO_FILL_UNIFORM = 1,    /* v_i = lo + (hi−lo)·U(seed ⊕ i), splitmix64 */
    CGO_FILL_ALTERNATE = 2   /* v_i = (i even) ? lo : hi   (Rosenbrock −1.2, 1) */
};

typedef struct cgo_beta_config {
    int32_t kind;
    int32_t lbfgs_m; /* history length for CGO_BETA_LBFGS */
    double mu;       /* YuanWangSheng μ, 0 < μ < 1 */
} cgo_beta_config;

/* CGConfig{T,BT,ET} (src/types.jl:156-169) built by setupCGConfig (:171-203) */
typedef struct cgo_cg_config {
    double eps;            /* ϵ, asserted 0 < ϵ < 1 (types.jl:187) */
    cgo_beta_config beta;  /* β_config */
    int64_t max_iters;
    int32_t verbose;       /* stored, not functional (as in the reference) */
    int32_t trace_enabled; /* EnableTrace / DisableTrace (types.jl:9-11) */
} cgo_cg_config;

/* union of StrongWolfeBisection (nocedal.jl:3-30),
 * WolfeBisection{Wolfe|YuanWeiLuWolfe} (wolfe.jl:6-11,213-217,259-262) and
 * Backtracking{Armijo} (geometric.jl:15-20,159-162) */
typedef struct cgo_ls_config {
    int32_t kind;
    int32_t cond_kind;
    double c1, c2;
    double a_max_growth_factor;
    double delta1;
    double max_step_size;
    int64_t max_iters;
    int64_t zoom_max_iters;
    int64_t feasibility_max_iters;
    double discount_factor;          /* Backtracking.discount_factor */
} cgo_ls_config;

/* Results{T,TrT} (src/types.jl:107-114) + TraceContainer{T,ET} (:17-23).
 * All pointers are caller-allocated host buffers and may be NULL to skip. */
typedef struct cgo_results {
    double objective;
    double *minimizer;              /* [n_local] this rank's shard */
    double *gradient;               /* [n_local] */
    int64_t iters_ran;
    int32_t status;
    int32_t _pad;
    double *trace_objective;        /* [max_iters]; valid [0, iters_ran) */
    double *trace_grad_norm;
    double *trace_step_size;
    int64_t *trace_objective_evals;
    int64_t total_fdf_evals;        /* engine counter: objective evaluations incl. the initial one */
    int64_t total_launches;         /* engine counter: kernel launches */
} cgo_results;

typedef struct cgo_ctx cgo_ctx;
typedef struct cgo_objective cgo_objective;
typedef struct cgo_solver cgo_solver;

/* cross-rank exchange hook for hosts that bring their own communicator
 * (e.g. MPI.jl): gather `count` doubles from every rank, rank-major, into
 * recv[world*count].  Must return 0 on success. */
typedef int (*cgo_allgather_fn)(void *user, const double *send, double *recv, int32_t count);

/* ---- library ----------------------------------------------------------- */
int cgo_version(void);
/* hex digest of the sources this binary was built from (the .hip / .hpp files of csrc and this header): lets a
 * measurement (the PMC summaries under profiles/) be tied to the exact kernels it was taken on */
const char *cgo_build_id(void);
const char *cgo_last_error(void);
const char *cgo_status_name(int32_t status);   /* the reference's Symbol text */
int cgo_device_count(int32_t *count);

/* config validation = the reference's @assert sites:
 * types.jl:187; nocedal.jl:22-26; wolfe.jl:233,278; geometric.jl:169 */
int cgo_check_cg_config(const cgo_cg_config *cfg);
int cgo_check_ls_config(const cgo_ls_config *ls);

/* ---- context: one GPU, one shard --------------------------------------- */
int cgo_ctx_create(int32_t device, cgo_ctx **out);
int cgo_ctx_destroy(cgo_ctx *ctx);
/* RCCL communicator over xGMI for the scalar exchange. unique_id: 128 bytes
 * from cgo_comm_unique_id on rank 0, broadcast by the host. */
int cgo_comm_unique_id(void *out128);
int cgo_ctx_set_comm_rccl(cgo_ctx *ctx, int32_t rank, int32_t world, const void *unique_id128);
int cgo_ctx_set_comm_callback(cgo_ctx *ctx, int32_t rank, int32_t world, cgo_allgather_fn fn,
                              void *user);
/* Single-node, lowest latency: a POSIX shared-memory mailbox.  Rank 0 calls with create = 1 first,
 * the host synchronises, the other ranks call with create = 0, the host synchronises again and rank
 * 0 may cgo_shm_unlink(name).  Every rank's finalize kernel then stores its ≤ 64-double block and a
 * sequence word directly into its slot of the segment (registered with HIP); all ranks read all
 * slots from host memory: no collective call per launch. */
int cgo_ctx_set_comm_shm(cgo_ctx *ctx, int32_t rank, int32_t world, const char *name, int32_t create);
int cgo_shm_unlink(const char *name);
/* Device mailboxes for the shared-memory communicator (SURVEY.md §8(e) "fast path"): COLLECTIVE — every rank calls it once,
 * after all ranks have attached to the segment.  Each rank exported a small block of its own HBM through the segment
 * (hipIpcGetMemHandle); this call opens every peer's (hipIpcOpenMemHandle: over xGMI between GPUs of a node) and agrees
 * with the others on whether everybody could.  *connected = 1: the finisher of a controller-armed launch now stores its
 * block straight into its peers' GPUs and sums the world's blocks itself, so armed rounds (csrc/cgo_ctl.hpp) work across
 * ranks without the host; host-driven launches keep publishing into the host slots.  *connected = 0 (more than 8 ranks, IPC
 * refused, CGO_NO_DEVICE_MAILBOX set): nothing changes.  Never an error for that reason. */
int cgo_ctx_comm_connect_devices(cgo_ctx *ctx, int32_t *connected);
/* Diagnostics of the scalar exchange (the reference has no counterpart: SURVEY.md §5 "Distributed
 * communication backend: none").  kind: 0 = single rank, 1 = shared-memory mailbox, 2 = RCCL, 3 = host callback.
 * ranks_seen: how many ranks the transport itself reports — ncclCommCount for RCCL, the number of mailbox slots
 * that have published at least one launch for the mailbox, `world` for a callback. */
int cgo_ctx_comm_info(cgo_ctx *ctx, int32_t *kind, int32_t *rank, int32_t *world, int32_t *ranks_seen);
/* Exchange cost since the last reset.  exchanges: launches whose sums crossed ranks.  peer_wait_us: host time
 * between this rank's own block being visible and the LAST peer's block being visible (mailbox: skew + PCIe
 * latency; 0 for the other transports).  device_exchange_us: mean duration on the GPU of the all-gather +
 * publish of a sampled launch (RCCL and forced-gather paths; HIP events around every 4th exchange; 0 otherwise).
 * reset != 0 zeroes the counters after reading them. */
int cgo_ctx_exchange_stats(cgo_ctx *ctx, int64_t *exchanges, double *peer_wait_us, double *device_exchange_us,
                           int32_t reset);
/* 1 if librccl can be loaded in this process (cgo_ctx_set_comm_rccl is a collective call: check first) */
int cgo_rccl_available(void);

/* ---- objective descriptor ---------------------------------------------- */
int cgo_objective_create(cgo_ctx *ctx, int32_t kind, int64_t n_global, int64_t offset,
                         int64_t n_local, cgo_objective **out);
/* "User-supplied element-wise f/∇f": the device-side form of handing `minimizeobjective` an
 * arbitrary `fdf!` closure (src/engine/optim.jl:25).  `source` is HIP C++ text compiled at run
 * time (hiprtc, gfx950, -ffp-contract=off) into the same fused kernel templates the built-in
 * objectives use.  Either
 *   - the statements of an element-wise body that set `fi` (the term f_i) and `gi` (∂f_i/∂x_i)
 *     from `x` (the element), `p` (its entry of parameter vector slot 0, if has_param) and `s0`
 *     (scalar slot 0), e.g.  "gi = p*x; fi = 0.5*(gi*x);"   or
 *   - a complete `struct UserObjective { kParam; kPairOnly; eval2(); eval1(); };` (the functor
 *     interface of csrc/cgo_kernels.hip.hpp) for objectives coupling the two elements of a pair.
 * On a compile error returns CGO_EINVAL; cgo_last_error() holds the hiprtc log. */
int cgo_objective_create_from_source(cgo_ctx *ctx, const char *source, int32_t has_param,
                                     int64_t n_global, int64_t offset, int64_t n_local,
                                     cgo_objective **out);
/* The reference's objective contract itself: `f = fdf!(g, x)` (src/engine/optim.jl:25, src/cg_utils.jl:19; exemplar
 * examples/helpers/test_funcs.jl:3-12) as a C callback — what Julia's `@cfunction` of an existing `fdf!` closure
 * produces, so that `minimizeobjective(boothfdf!, x0, config, ls)` (examples/min.jl:41) is a drop-in without rewriting
 * the objective as device code.  This is still the GPU engine: x, u, both gradients, every AXPY / direction update /
 * dot / norm / getβ sum and the line-search state machine stay where they are for the built-in objectives; per trial the
 * engine forms xp = x + a·u on the device straight into pinned host memory, calls `fn` there (it writes g, returns f)
 * and moves g back.  One trial step per launch, no on-device controller; CG β kinds, L-BFGS, all three line searches.
 * The callback sees this rank's shard [offset, offset + n_local) and returns this rank's part of f (the parts are summed
 * across ranks) — only separable objectives shard.  Throughput is PCIe-bound (16 B/element per trial): the device
 * objectives above are the fast path. */
typedef double (*cgo_fdf_fn)(void *user, double *g_local, const double *x_local, int64_t n_local);
int cgo_objective_create_callback(cgo_ctx *ctx, cgo_fdf_fn fn, void *user, int64_t n_global, int64_t offset,
                                  int64_t n_local, cgo_objective **out);
int cgo_objective_destroy(cgo_objective *obj);
int cgo_objective_set_param_host(cgo_objective *obj, int32_t slot, const double *host_local);
int cgo_objective_fill_param(cgo_objective *obj, int32_t slot, int32_t fill_kind, uint64_t seed,
                             double lo, double hi);
int cgo_objective_set_scalar(cgo_objective *obj, int32_t slot, double value);
/* Cost class of a user objective: how many speculative trial steps a fused launch evaluates.
 * 0 (default for cgo_objective_create_from_source): heavier than a few flops per element — one step per launch
 * below n_local = 3e6, three above, on-device controller on.  1: cheap (f and ∇f cost ≲ 10 flops per element,
 * like the built-in quadratic) — seven steps per launch.  Built-in objectives carry their own class. */
int cgo_objective_set_cost_class(cgo_objective *obj, int32_t cost_class);
/* U2: f = fdf!(g, x) for host vectors (H2D, one launch, D2H) — KAT entry */
int cgo_objective_eval_host(cgo_objective *obj, const double *x_local, double *g_local,
                            double *f_global);

/* ---- solver policy -------------------------------------------------------
 * HOW a solve runs — never WHAT it computes: every setting below changes launch counts, kernels or hand-off protocols,
 * not the step sequence (the parity tests hold each of them to the same oracle).  A field left at its "library policy"
 * value (what cgo_solver_policy_init writes) is decided by the library from the objective, the problem size and the
 * context; the CGO_* environment variables of earlier rounds remain as overrides FOR EXPERIMENTS and apply only to fields
 * left at "library policy" (an explicit field always wins).  No counterpart in the reference (it has one code path).
 * Order of precedence per field: cgo_solver_create_ex argument > cgo_ctx_set_default_policy > environment > library. */
typedef struct cgo_solver_policy {
    int32_t size;                 /* sizeof(cgo_solver_policy), written by cgo_solver_policy_init (versioning) */
    int32_t points;               /* trial steps per fused launch: 0 library policy | 1 | 3 | 5 | 7   */
    int32_t resident;             /* resident solver (§2.12): -1 library policy | 0 off | 1 on where objective and size allow */
    int32_t controller_depth;     /* on-device line-search controller (§2.7): -1 library policy | 0 host-driven | k armed rounds in flight */
    int32_t controller_graph;     /* armed rounds replayed from hipGraphs: -1 library policy (off) | 0 | 1 */
    int32_t controller_fused;     /* an armed round as ONE launch: -1 library policy (on) | 0 reduce + controller launches | 1 */
    int32_t stored_gradient;      /* 0 library policy (gradient-free k_cg family where it applies) | 1 stored-gradient k_fused family */
    int32_t fused_tail;           /* a launch sums its own rows (§2.4): -1 keep the context's setting | 0 finalize launches | 1 fused.  CONTEXT-WIDE */
    int32_t strict_tail;          /* hand-offs by __threadfence_system + release store instead of self-validating blocks (§2.4):
                                     -1 keep the context's setting (off) | 0 | 1.  CONTEXT-WIDE: applies to later solvers of the ctx too */
    int32_t placement_search;     /* buffer placement search at pure-HBM sizes (§2.5; a measured HEURISTIC: it finds a faster
                                     (x, u, D) buffer triple in 5 of 8 processes, none in the others; 10–450 ms and up to
                                     placement_max_bytes of transient memory per solver): -1 library policy (OFF: opt-in) | 0 off | 1 on */
    int32_t placement_stages;     /* 0 library policy (3) | 1..3: stages of 8 spare vectors the search may allocate */
    int64_t placement_max_bytes;  /* cap on the search's TRANSIENT device memory: 0 library policy (min(24 vectors, a quarter of the
                                     free memory)) | bytes.  Below 8 vectors' worth the search does not run */
    int32_t lbfgs_form;           /* 0 library policy | 1 one ring pass per iteration, state update riding in the next pass (§2.3) |
                                     2 one ring pass + a state-update launch of its own | 3 Gram form, two passes | 4 chained two-loop */
    int32_t lbfgs_fuse_grad;      /* the log-sum-exp push forms g⁺ itself: -1 library policy (on) | 0 | 1 */
    int32_t lbfgs_fuse_trial;     /* first trial of the next line search in the direction pass: -1 library policy (on) | 0 | 1 */
    int32_t lse_fixed_reference;  /* log-sum-exp statistics against a fixed reference instead of a running maximum: -1 (on) | 0 | 1 */
    int32_t resident_points;      /* 0 library policy (3) | 1 | 3 | 7: trial steps per pass of the resident solver */
    int32_t resident_chunk;       /* 0 library policy (4096) | even number of elements per workgroup of the resident solver */
    double hbm_stream_bytes;      /* a launch moving more than this streams pure-HBM style (contiguous chunks, non-temporal; §2.5):
                                     0 library policy (1.4e9 read-write, 4.5e8 read-only) | bytes (1 = every launch) */
    int32_t reserved[8];          /* zero */
} cgo_solver_policy;
void cgo_solver_policy_init(cgo_solver_policy *p);                               /* every field = library policy */
int cgo_ctx_set_default_policy(cgo_ctx *ctx, const cgo_solver_policy *policy);   /* for every solver this context creates from now on,
                                                                                    the one-shot entry points included; NULL resets */
/* what a solver actually runs with, after the precedence above (reporting; tests) */
int cgo_solver_get_policy(cgo_solver *s, cgo_solver_policy *out);

/* ---- solver: resumable form of minimizeobjective (optim.jl:6-171) ------
 * Memory: x and u (16 B per local element; the k_cg family keeps no gradient vector) — plus two gradient buffers for the
 * stored-gradient families (quasi-Newton flavours, host closures, log-sum-exp), the 2(m + 1) vectors of the L-BFGS ring,
 * and a second iterate buffer for L-BFGS on log-sum-exp (the fused push advances x out of place).
 * Creation time: with cgo_solver_policy.placement_search = 1 (opt-in since round 4; bench.py opts in) and a pure-HBM problem
 * size (the dominant launch moves more than 1.4 GB: n_local ≳ 3.5e7 for the quadratic) creating a solver runs the placement
 * search of DESIGN.md §2.5: up to 24 spare n-vectors — never more than placement_max_bytes, by default a quarter of the free
 * device memory — are allocated transiently and up to ≈ 190 short launches timed — 10–150 ms at n = 1e8 (≤ 450 ms observed) —
 * and the parameter vector of the objective may be moved to another buffer (only while this solver is the objective's sole
 * user).  The pair (x, u) it kept is parked in the context when the solver is destroyed and taken over by the next solver of
 * the same size (rerun chains, centering steps); parked buffers of another size are released before a new solver allocates. */
int cgo_solver_create_ex(cgo_ctx *ctx, cgo_objective *obj, const cgo_cg_config *cfg, const cgo_ls_config *ls,
                         const cgo_solver_policy *policy /* NULL = the context's default */, cgo_solver **out);
int cgo_solver_create(cgo_ctx *ctx, cgo_objective *obj, const cgo_cg_config *cfg,
                      const cgo_ls_config *ls, cgo_solver **out);
int cgo_solver_destroy(cgo_solver *s);
int cgo_solver_set_x0_host(cgo_solver *s, const double *x0_local);
int cgo_solver_set_x0_fill(cgo_solver *s, int32_t fill_kind, uint64_t seed, double lo, double hi);
/* The same for callers whose vectors already live in THIS GPU's memory (a ROCArray, a torch tensor): x0_dev is a
 * device pointer to n_local doubles, copied device-to-device (optim.jl:21 copies x_initial too).  No PCIe traffic:
 * at n = 1e8 the host form moves 0.8 GB in (and cgo_solver_results 1.6 GB out) ≈ 50 ms, against 0.9 ms per iteration. */
int cgo_solver_set_x0_device(cgo_solver *s, const double *x0_dev);
/* optim.jl:25-47: initial fdf!, ‖g‖, u = −g */
int cgo_solver_start(cgo_solver *s);
/* optim.jl:50-160: run at most `iters` further outer iterations;
 * *finished = 1 once a terminal status was reached */
int cgo_solver_iterate(cgo_solver *s, int64_t iters, int32_t *finished);
/* optim.jl:162-170 / updateresult! (types.jl:134-151).  Also where a reduction that gave up on a partial-row slot
 * (it leaves a NaN in that launch's sums) surfaces: CGO_EHIP instead of a record that blames the objective. */
int cgo_solver_results(cgo_solver *s, cgo_results *out);
/* Results.minimizer / Results.gradient (types.jl:107-114) into DEVICE buffers of n_local doubles each (either may be
 * NULL); everything else of the record through cgo_solver_results with NULL vector pointers. */
int cgo_solver_results_device(cgo_solver *s, double *minimizer_dev, double *gradient_dev);
/* branch log of every evalϕdϕ! (cg_utils.jl:4-23): (a, ϕ, dϕ); returns count */
int cgo_solver_trial_log(cgo_solver *s, int64_t cap, double *a, double *phi, double *dphi,
                         int64_t *count);
/* per-kernel-kind HIP-event timings accumulated since start / last reset */
int cgo_solver_profile_enable(cgo_solver *s, int32_t on);
int cgo_solver_profile_reset(cgo_solver *s);
int cgo_solver_profile_get(cgo_solver *s, int32_t kernel_kind, int64_t *launches,
                           double *total_ms, double *bytes_per_launch);
const char *cgo_kernel_kind_name(int32_t kernel_kind);
/* which kernel family the solver launches: "k_cg (gradient-free, N-point)" with N = 1, 3, 5 or 7 trial steps per launch,
 * "k_fused (stored gradient)", "k_lse (two-phase)"; L-BFGS adds "+ k_lbfgs" */
const char *cgo_solver_kernel_family(cgo_solver *s);
/* Launches that were armed by the on-device controller (csrc/cgo_ctl.hpp) instead of the host:
 * streaks of outer iterations whose line search accepts its first trial (nocedal.jl:78-110,
 * wolfe.jl:51-78) run device-side; the host replays them from published records.  Depth of the
 * run-ahead: env CGO_CTL_DEPTH (0 = the host drives every launch; default 4 for objectives whose
 * launches carry at most three trial steps, 0 for the cheap built-in ones — DESIGN.md §2.7). */
int64_t cgo_solver_controller_launches(cgo_solver *s);
/* Resident solver (csrc/cgo_resident.hpp): for cache-sized shards — x, u and the parameter vector fit the LDS of the chip,
 * n ≲ 1.6e6 with a parameter vector — of the built-in element-wise objectives under a CGβConfig and StrongWolfeBisection /
 * WolfeBisection on one rank, cgo_solver_iterate runs a whole slice of outer iterations (optim.jl:50-160: line search,
 * getβ, updatedir!) inside ONE launch; iterations it cannot complete (any outcome other than :success, rare-path norms)
 * are run by the host-driven path as before.  Reports the slices launched, the iterations completed inside them, and how
 * often a slice was given up because its workgroups could not all run at once (another process holding CUs): such a slice
 * changes nothing, the launch-per-trial engine redoes it, and the solver stays off the resident path afterwards.
 * Environment: CGO_RESIDENT=0 switches it off; CGO_RES_CHUNK (elements per workgroup), CGO_RES_POINTS (1 | 3 | 7). */
int cgo_solver_resident_stats(cgo_solver *s, int64_t *slices, int64_t *iterations, int64_t *gave_up);
/* L-BFGS (Gram form, m ≤ 10) on the log-sum-exp, separable-quadratic, paired-Rosenbrock and user-compiled objectives: how the state updates
 * ("pushes") of this solver were paid for.
 * `speculated`: the direction pass had already taken every inner product at the step that was then accepted — one pass
 * over the ring for that iteration (k_lbfgs_combine_spec; the 56 B/element state update rides in the NEXT direction
 * pass, or runs as k_lbfgs_push_lite when none follows); `fused`: the push read the ring and formed g⁺ itself
 * (k_lbfgs_push_gram_lse; log-sum-exp only); `plain`: k_lbfgs_push_gram on a gradient a launch of its own wrote.
 * Environment: CGO_LBFGS_SPEC=0 (1: the state update always as its own launch), CGO_LBFGS_FUSE_GRAD=0 switch the first two off. */
int cgo_solver_lbfgs_stats(cgo_solver *s, int64_t *speculated, int64_t *fused, int64_t *plain);
int cgo_num_kernel_kinds(void);
/* The kernel instantiation a launch of `kernel_kind` uses under the solver's current policy, as the profiler
 * prints it without namespaces — e.g. "k_cg<ObjQuadDiag, 7, 7, true>" (objective, mode bits, trial points, pure-HBM
 * streaming policy).  Written NUL-terminated into buf[cap]. */
int cgo_solver_kernel_symbol(cgo_solver *s, int32_t kernel_kind, char *buf, int32_t cap);

/* ---- the line-search conditions as scalar functions ------------------------------------------------
 * evalwolfeconditions(condition, ϕ_a, dϕ_a, a, u, ϕ_0, dϕ_0) → (valid_large, valid_small) — wolfe.jl:219-294
 * (Wolfe: :264-294, YuanWeiLuWolfe: :219-251; `uu` = dot(u,u), which the YWL form reads at :240) — and
 * evalbacktrackcondition(::Armijo, ϕ_a, a, ϕ_0, dϕ_0) — geometric.jl:164-186.  The same code the engine and
 * the on-device controller run (csrc/cgo_ctl.hpp).  `ls` supplies cond_kind, c1, c2, delta1. */
int cgo_evalwolfeconditions(const cgo_ls_config *ls, double phi_a, double dphi_a, double a, double uu,
                            double phi_0, double dphi_0, int32_t *valid_large, int32_t *valid_small);
int cgo_evalbacktrackcondition(const cgo_ls_config *ls, double phi_a, double a, double phi_0, double dphi_0,
                               int32_t *valid);

/* ---- solvesystem (src/engine/solve_system.jl; exported at ConjugateGradientOptim.jl:28) ---- */
/* LinesearchSolveSys{T} + setupLinesearchSolveSys — solve_system.jl:6-27 (eqn 18 of Yuan 2019) */
typedef struct {
    double rho;        /* 0 < ρ < 1 (asserted, solve_system.jl:21-22): step shrink factor */
    double sigma;      /* σ (default 0.5; the reference never checks it) */
    double s;          /* s > 0 (asserted, :23): the first trial step */
    int64_t max_iters; /* default round(Int, log(ρ, 1e-6)) = cgo_lss_default_max_iters(ρ) */
} cgo_lss_config;
int cgo_check_lss_config(const cgo_lss_config *ls);
int64_t cgo_lss_default_max_iters(double rho);
/* A solver running solvesystem(fdf!, x_initial, config, linesearch_config) — solve_system.jl:64-237 —
 * instead of minimizeobjective; every other cgo_solver_* call applies unchanged.  CG β kinds and
 * element-wise objectives only.  Restated bug for bug: the projection step is
 * added to the x_next buffer, i.e. to the iterate of two iterations ago (:172-178,:194);
 * trace.objective_evals holds the 0-based index of the accepted trial (:52); status
 * CGO_LINESEARCH_FAILED marks the point where the reference throws UndefVarError (:55). */
int cgo_solver_create_sys(cgo_ctx *ctx, cgo_objective *obj, const cgo_cg_config *cfg,
                          const cgo_lss_config *ls, cgo_solver **out);
int cgo_solver_create_sys_ex(cgo_ctx *ctx, cgo_objective *obj, const cgo_cg_config *cfg, const cgo_lss_config *ls,
                             const cgo_solver_policy *policy /* NULL = the context's default */, cgo_solver **out);
int cgo_solvesystem(cgo_ctx *ctx, cgo_objective *obj, const double *x0_local,
                    const cgo_cg_config *cfg, const cgo_lss_config *ls, cgo_results *out);

/* ---- one-shot drop-ins -------------------------------------------------- */
/* minimizeobjective(fdf!, x_initial, config, linesearch_config)  optim.jl:6-11 */
int cgo_minimize(cgo_ctx *ctx, cgo_objective *obj, const double *x0_local,
                 const cgo_cg_config *cfg, const cgo_ls_config *ls, cgo_results *out);
/* minimizeobjectivererun(fdf!, x_initial, config, ls, rerun_config_tuples...)
 * optim.jl:173-208.  outs has capacity 1 + npairs; *nouts = runs performed.  A stage restarts from the previous
 * stage's minimizer (optim.jl:195-200), which stays on the GPU between the stages (device-to-device copy): the
 * minimizer / gradient buffers of ANY outs[k] may be NULL where the host does not want that stage's vectors.
 * Sharded contexts: every rank calls this with its own shard; all ranks see the same statuses and stage count. */
int cgo_minimize_rerun(cgo_ctx *ctx, cgo_objective *obj, const double *x0_local,
                       const cgo_cg_config *cfg, const cgo_ls_config *ls,
                       const cgo_cg_config *rerun_cfgs, const cgo_ls_config *rerun_ls,
                       int32_t npairs, cgo_results *outs, int32_t *nouts);

/* ---- kernel-level entry points on host vectors (KATs / micro-benchmarks)
 *      each: H2D, ONE fused launch, D2H ---------------------------------- */
/* updatedir!(u, df_x, β) (cg_flavours.jl:2-15) fused with the next dϕ₀ = g·u
 * (nocedal.jl:56, wolfe.jl:40) and u·u (wolfe.jl:240): out2 = {g·u_new, u_new·u_new} */
int cgo_kernel_dir(cgo_ctx *ctx, double *u, const double *g, double beta, int64_t n,
                   double *out2);
/* one-pass partial sums for getβ (cg_flavours.jl:46-170):
 * out9 = {g⁺·u, g⁺·g⁺, g⁺·g, y·y, u·y, y·g⁺, g·g, g·u, u·u}, y = g⁺ − g */
int cgo_kernel_beta_partials(cgo_ctx *ctx, const double *g_next, const double *g,
                             const double *u, int64_t n, double *out9);
/* getβ(β_config, g_next, g, u)::T evaluated from those partials (scalar work) */
int cgo_getbeta(cgo_ctx *ctx, const cgo_beta_config *b, const double *g_next, const double *g,
                const double *u, int64_t n, double *beta);
/* evalϕdϕ!(xp, df_xp, fdf!, a, x, u) (cg_utils.jl:4-23): out2 = {ϕ, dϕ}; g_next_out = df_xp */
int cgo_kernel_trial(cgo_objective *obj, const double *x, const double *u, double a,
                     double *g_next_out, double *out2);
/* device-resident micro-benchmark of the fused kernels: allocates vectors of
 * n doubles on the ctx, runs `reps` launches of `kernel_kind`, returns the mean
 * HIP-event time per launch (ms) and the algorithmic bytes per launch */
int cgo_bench_kernel(cgo_ctx *ctx, cgo_objective *obj, int32_t kernel_kind, int64_t n,
                     int32_t reps, double *ms_per_launch, double *bytes_per_launch);

/* What the box at hand delivers for the READ/WRITE MIX of the dominant launch (accept + direction + trial: R x, u, D /
 * W x, u in place, 40 B per element) with next to no arithmetic, under the engine's pure-HBM streaming policy: median
 * and best HIP-event time of `reps` launches on n doubles per vector.  bench.py reports the engine's launch against this
 * measured ceiling beside the 8 TB/s pin peak. */
int cgo_bench_stream_mix(cgo_ctx *ctx, int64_t n, int32_t reps, double *median_us, double *best_us);
/* Placement search of a solver on a pure-HBM problem size: the time of that bare mix on the buffers as
 * first allocated, on the triple (x, u, D) the solver kept, and how many candidate triples were timed (0 = no search:
 * problem below the pure-HBM threshold, CGO_PLACE_TUNE=0, or not enough free memory for the spare buffers). */
int cgo_solver_placement_info(cgo_solver *s, double *as_allocated_us, double *chosen_us, int32_t *candidates);

#ifdef __cplusplus
}
#endif
#endif /* CGO_H */
