/*
 * cgo_oracle.h — CPU ORACLE for the nonlinear-CG / quasi-Newton hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker / the timed CPU baseline.  Nothing under
 * conjugategradientoptim.jl_amd/ links, imports or executes it.
 *
 * It is a plain-C restatement, pass for pass, of the reference's Julia code
 * (RWAlgorithms/ConjugateGradientOptim.jl, paths relative to /root/reference):
 *   src/engine/optim.jl:6-208        minimizeobjective, minimizeobjectivererun
 *   src/cg_utils.jl:4-23             evalϕdϕ!
 *   src/cg_flavours.jl:2-170         updatedir!, initialize*, getβ ×4
 *   src/linesearch/nocedal.jl:3-209  StrongWolfeBisection, linesearch!, zoom!
 *   src/linesearch/wolfe.jl:6-294    WolfeBisection, findfeasiblestepsize!, Wolfe, YuanWeiLuWolfe
 *   src/types.jl:17-203              TraceContainer, Results, CGConfig
 *   src/linesearch/geometric.jl:15-186  Backtracking, Armijo
 *   src/engine/solve_system.jl:6-253 LinesearchSolveSys, linesearch!, solvesystem, updateiteratesolvesys!
 *
 * PARITY PINNING: the reference is Julia and cannot be executed in this
 * environment (no julia binary, no network), and its own test-suite
 * (test/runtests.jl:7-44) pins only the Booth gradient, never the solver.  The
 * oracle is therefore pinned by (i) that Booth known answer, (ii) hand-derived
 * scalar KATs of every β formula / Wolfe inequality / the first Booth iteration
 * (SURVEY.md appendix A), (iii) an independently written numpy restatement
 * (oracle/cgo_oracle_np.py) that must agree with this file, and (iv) the
 * closed-form linear-CG cross-check.  Solver trajectories are otherwise
 * "parity unpinned" by the reference itself.
 *
 * β flavours PR / HS / DY and L-BFGS do not exist in the reference (PRP is an
 * empty stub at cg_flavours.jl:173-174, HS is commented out at :110-127); they
 * are defined HERE first, behind the reference's getβ/updatedir! contract.
 */
#ifndef CGO_ORACLE_H
#define CGO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* U2: the reference's objective callback contract  f = fdf!(g, x)
 * (src/cg_utils.jl:19, src/engine/optim.jl:25).  Writes ∇f(x) into g. */
typedef double (*orc_fdf_t)(void *user, double *g, const double *x, int64_t n);

/* Status symbols of the reference, one integer per Symbol. */
enum {
    ORC_INCOMPLETE = 0,                                 /* optim.jl:39   */
    ORC_SUCCESS = 1,                                    /* optim.jl:64   */
    ORC_INCREASING_OBJECTIVE = 2,                       /* optim.jl:76   */
    ORC_NON_FINITE_OBJECTIVE_OR_GRADIENT_PROPOSED = 3,  /* optim.jl:118  */
    ORC_MAX_ITERS_REACHED = 4,                          /* optim.jl:168  */
    ORC_NON_DESCENT_SEARCH_DIRECTION = 5,               /* nocedal.jl:62, wolfe.jl:42 */
    ORC_LINESEARCH_A_MAX_OVERFLOW = 6,                  /* nocedal.jl:148 */
    ORC_LINESEARCH_MAX_ITERS_REACHED = 7,               /* nocedal.jl:157, wolfe.jl:164 */
    ORC_ZOOM_MAX_ITERS_REACHED = 8,                     /* nocedal.jl:208 */
    ORC_ACCEPTED_NON_FINITE_ITERATE = 9,                /* wolfe.jl:37   */
    ORC_CANNOT_FIND_INITIAL_FEASIBLE_STEP = 10,         /* wolfe.jl:64   */
    ORC_MAX_STEP_LENGTH_REACHED = 11,                   /* wolfe.jl:111  */
    ORC_CANNOT_FIND_FEASIBLE_STEP = 12,                 /* wolfe.jl:157  */
    ORC_STEP_BRACKET_PRECISION_ISSUE = 13,              /* wolfe.jl:131 (dead: missing return) */
    ORC_BISECTION_LOWER_BOUND_LARGER_THAN_PROPOSED_STEP = 14, /* wolfe.jl:187 */
    ORC_FEASIBLE = 15,                                  /* wolfe.jl:197  */
    ORC_INFEASIBLE = 16,                                /* wolfe.jl:206  */
    ORC_NON_FINITE_STEP_PROPOSED = 17,                  /* geometric.jl:129 */
    ORC_PROPOSED_STEP_SAME_AS_CURRENT_STEP = 18,        /* geometric.jl:133 */
    ORC_LINESEARCH_FAILED = 19,                         /* solve_system.jl:139 (see orc_solvesystem) */
    ORC_NUM_STATUS = 20
};

enum { /* βConfig subtypes (cg_flavours.jl) */
    ORC_BETA_HAGER_ZHANG = 0,       /* cg_flavours.jl:83-108  */
    ORC_BETA_YUAN_WANG_SHENG = 1,   /* cg_flavours.jl:46-79   */
    ORC_BETA_SALLEH_ALHAWARAT = 2,  /* cg_flavours.jl:130-151 */
    ORC_BETA_LIU_STORREY = 3,       /* cg_flavours.jl:154-170 */
    ORC_BETA_POLAK_RIBIERE = 4,     /* NEW (stub at cg_flavours.jl:173-174) */
    ORC_BETA_HESTENES_STIEFEL = 5,  /* NEW (commented at cg_flavours.jl:110-127) */
    ORC_BETA_DAI_YUAN = 6,          /* NEW */
    ORC_BETA_LBFGS = 7,             /* NEW QNβConfig (contract: qn_flavours.jl:5-48) */
    ORC_BETA_BROYDEN_FAMILY = 8     /* qn_flavours.jl:53-90, dense n×n as written (θ in the mu field); n ≤ 4096 */
};

enum { ORC_LS_STRONG_WOLFE_BISECTION = 0, /* nocedal.jl */
       ORC_LS_WOLFE_BISECTION = 1,        /* wolfe.jl   */
       ORC_LS_BACKTRACKING = 2 };         /* geometric.jl:15-152 */
enum { ORC_COND_WOLFE = 0,                /* wolfe.jl:259-294 */
       ORC_COND_YUAN_WEI_LU = 1,          /* wolfe.jl:213-251 */
       ORC_COND_ARMIJO = 2 };             /* geometric.jl:159-186 */

typedef struct {
    int32_t kind;
    int32_t lbfgs_m;   /* LBFGS history length */
    double mu;         /* YuanWangSheng μ */
} orc_beta_config;

typedef struct {
    int32_t kind;
    int32_t cond_kind;               /* WolfeBisection.condition */
    double c1, c2;
    double a_max_growth_factor;      /* StrongWolfeBisection (nocedal.jl:8) */
    double delta1;                   /* YuanWeiLuWolfe.δ1 */
    double max_step_size;            /* WolfeBisection (wolfe.jl:9) */
    int64_t max_iters;
    int64_t zoom_max_iters;          /* nocedal.jl:10 */
    int64_t feasibility_max_iters;   /* wolfe.jl:10, geometric.jl:19 */
    double discount_factor;          /* Backtracking (geometric.jl:17) */
} orc_ls_config;

typedef struct {           /* CGConfig (types.jl:156-169) */
    double eps;            /* ϵ */
    orc_beta_config beta;
    int64_t max_iters;
    int32_t verbose;
    int32_t trace_enabled; /* EnableTrace / DisableTrace */
} orc_cg_config;

typedef struct {           /* Results + TraceContainer (types.jl:17-23,107-114) */
    double objective;
    double *minimizer;     /* caller-allocated [n] */
    double *gradient;      /* caller-allocated [n] */
    int64_t iters_ran;
    int32_t status;
    int32_t _pad;
    double *trace_objective;        /* caller-allocated [max_iters] or NULL */
    double *trace_grad_norm;
    double *trace_step_size;
    int64_t *trace_objective_evals;
    /* branch log of every evalϕdϕ! call (oracle extension for divergence triage) */
    int64_t log_cap, log_len;
    double *log_a, *log_phi, *log_dphi;
    int64_t total_fdf_evals;        /* includes the initial fdf! at optim.jl:25 */
    /* [log_cap] or NULL: per logged evaluation, the smallest relative margin |lhs − rhs| / scale of the line-search
     * branches decided on it (nocedal.jl:81-112,192-200, wolfe.jl:80-122 via :286-291/:243-248, geometric.jl:80,139) */
    double *log_margin;
    /* checkpoints (oracle extension for the long-horizon error curves): the iterate x after outer iteration
     * snap_iters[j] (1-based, ascending) is copied to snap_x + j·n; snap_done = how many were reached */
    int64_t nsnap, snap_done;
    const int64_t *snap_iters;
    double *snap_x;
    /* [max_iters] or NULL: CLOCK_MONOTONIC seconds at the end of every outer iteration (bench.py's cpu_baseline times
     * iterations w+1 … w+k of ONE run at the full problem size from these: no differencing of runs, no extrapolation) */
    double *trace_time;
} orc_results;

const char *orc_status_name(int status);

/* config validation = the reference's @assert sites; 0 = ok, else line-coded error */
int orc_check_cg_config(const orc_cg_config *cfg);   /* types.jl:187 */
int orc_check_ls_config(const orc_ls_config *ls);    /* nocedal.jl:22-26, wolfe.jl:233,278 */

/* optim.jl:6-171 */
int orc_minimizeobjective(orc_fdf_t fdf, void *user, const double *x_initial, int64_t n,
                          const orc_cg_config *cfg, const orc_ls_config *ls, orc_results *ret);

/* optim.jl:173-208; rets[0..*nrets) filled, rets has capacity 1+npairs */
int orc_minimizeobjectivererun(orc_fdf_t fdf, void *user, const double *x_initial, int64_t n,
                               const orc_cg_config *cfg, const orc_ls_config *ls,
                               const orc_cg_config *rerun_cfgs, const orc_ls_config *rerun_ls,
                               int npairs, orc_results *rets, int *nrets);

/* solve_system.jl:6-27  LinesearchSolveSys / setupLinesearchSolveSys (eqn 18 of Yuan 2019) */
typedef struct {
    double rho;        /* 0 < ρ < 1 (asserted :21-22) */
    double sigma;      /* σ (default 0.5; NOT asserted by the reference) */
    double s;          /* s > 0 (asserted :23): first trial step */
    int64_t max_iters; /* default round(Int, log(ρ, 1e-6)) */
} orc_lss_config;
int orc_check_lss_config(const orc_lss_config *l);      /* solve_system.jl:21-23 */
int64_t orc_lss_default_max_iters(double rho);          /* solve_system.jl:17 */

/* solve_system.jl:64-253  solvesystem (+ linesearch! :29-56, updateiteratesolvesys! :239-253),
 * restated bug for bug:
 *  - updateiteratesolvesys! adds the projection step to `x_next`, which after the first swap holds
 *    the iterate of TWO iterations ago, not the current one (:172-178, :199);
 *  - trace.objective_evals receives the 0-based index i of the accepted trial, i.e. evals − 1 (:52);
 *  - when no trial passes, linesearch! reads the loop variable `i` outside its scope (:55): Julia
 *    throws UndefVarError there, so :linesearch_failed (:131-141) is unreachable in the reference.
 *    The oracle returns ORC_LINESEARCH_FAILED with the last good iterate (the evident intent) and
 *    sets ret->_pad = 1 to flag "the reference throws here". */
int orc_solvesystem(orc_fdf_t fdf, void *user, const double *x_initial, int64_t n,
                    const orc_cg_config *cfg, const orc_lss_config *ls, orc_results *ret);

/* --- scalar-level entry points for KATs -------------------------------- */
double orc_getbeta(const orc_beta_config *b, const double *g_next, const double *g,
                   const double *u, int64_t n);                       /* cg_flavours.jl getβ */
void orc_updatedir(double *u, const double *df_x, double beta, int64_t n); /* cg_flavours.jl:2-15 */
void orc_evalwolfeconditions(const orc_ls_config *ls, double phi_a, double dphi_a, double a,
                             const double *u, int64_t n, double phi_0, double dphi_0,
                             int *valid_large, int *valid_small);     /* wolfe.jl:219-294 */
double orc_dot(const double *a, const double *b, int64_t n);
double orc_norm(const double *a, int64_t n);
/* 1 in the arbiter build (-DORC_EXACT_SUMS: every reduction in twice the working precision, rounded once), else 0 */
int orc_exact_sums(void);
#ifdef ORC_EXACT_SUMS
double orc_sum(const double *v, int64_t n);
double orc_dot_f128(const double *a, const double *b, int64_t n);   /* __float128 accumulation: the pin of Dot2 */
double orc_sum_f128(const double *v, int64_t n);
#endif

/* --- built-in objectives (U2 contract) ---------------------------------- */
typedef struct { const double *D; } orc_quad_params;     /* f = ½ Σ D_i x_i²            */
typedef struct { double lambda; } orc_lse_params;        /* f = log Σ e^{x_i} + ½λ‖x‖²   */
double orc_fdf_booth(void *user, double *g, const double *x, int64_t n);          /* test_funcs.jl:3-12 */
double orc_fdf_quad_diag(void *user, double *g, const double *x, int64_t n);
double orc_fdf_rosenbrock_paired(void *user, double *g, const double *x, int64_t n);
double orc_fdf_rosenbrock_chained(void *user, double *g, const double *x, int64_t n); /* value: test_funcs.jl:50-57 */
double orc_fdf_lse(void *user, double *g, const double *x, int64_t n);

/* counter-based RNG shared with the device fill kernels: U(i) ∈ [0,1) */
double orc_uniform(uint64_t seed, uint64_t index);
void orc_fill_quad_diag(double *D, int64_t offset, int64_t n, uint64_t seed, double lo, double hi);
void orc_fill_uniform(double *v, int64_t offset, int64_t n, uint64_t seed, double lo, double hi);

#ifdef __cplusplus
}
#endif
#endif
