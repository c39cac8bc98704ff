/*
 * cgo_oracle.c — CPU ORACLE (test infrastructure; see cgo_oracle.h header).
 *
 * Plain-C, single-threaded, pass-for-pass restatement of the reference's
 * Julia hot path.  Every function cites the reference file:line it follows.
 * Deliberately keeps the reference's pass structure (separate AXPY / fdf! /
 * dot / norm passes, the getβ temporaries, the three per-iteration copies) so
 * that (a) rounding behaviour matches the reference's formulas, not the fused
 * GPU formulas, and (b) timing it gives the "reference-equivalent CPU path".
 *
 * Julia semantics honoured here:
 *   - no FMA contraction: Julia never fuses a*b+c unless asked → build with
 *     -ffp-contract=off;
 *   - max/min propagate NaN (cg_flavours.jl:68, wolfe.jl:243,247);
 *   - comparisons with NaN are false; `!(0 < a && isfinite(a))` style guards
 *     are restated literally;
 *   - isapprox(x, 0) with default tolerances is exact equality with 0
 *     (wolfe.jl:123);
 *   - norm(v) = sqrt(Σ v_i²) (LinearAlgebra.generic_norm2 fast path /
 *     BLAS.nrm2 — equal to rounding); dot = BLAS ddot, unspecified
 *     summation order → reference results are themselves defined only up to
 *     reduction-order noise, hence the 1e-10 relative parity bar.
 */
#define _POSIX_C_SOURCE 200809L   /* clock_gettime under -std=c11 */
#include "cgo_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double orc_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Built three times from this one source:
 *   _build/libcgo_oracle.so        plain: the parity oracle, strictly single-threaded;
 *   _build/libcgo_oracle_omp.so    -fopenmp: bench.py's all-cores CPU baseline and the BASELINE-size parity children
 *                                  (reductions = `omp parallel for reduction(+)`: another valid summation order);
 *   _build/libcgo_oracle_exact.so  -fopenmp -DORC_EXACT_SUMS: the ARBITER.  Every reduction of the path — dot, norm,
 *                                  the objective sums — is accumulated in twice the working precision (Ogita, Rump &
 *                                  Oishi's Dot2 / Sum2 on fixed 65 536-element blocks, the block results added in
 *                                  double-double) and rounded to Float64 ONCE: the value every summation order of the
 *                                  reference's `LinearAlgebra.dot` / `norm` approximates.  Everything else — the
 *                                  element-wise vector arithmetic and every scalar formula of the line searches and of
 *                                  getβ — stays IEEE double exactly as the reference writes it (those are specified by
 *                                  the Julia source; only the order of a reduction is not).  The result does not depend
 *                                  on the thread count.  orc_dot_f128 (__float128 accumulation) pins Dot2 in the tests.
 */
#define ORC_OMP_MIN 100000
#define ORC_PRAGMA(x) _Pragma(#x)
#ifdef _OPENMP
#define ORC_PAR _Pragma("omp parallel for schedule(static) if (n >= ORC_OMP_MIN)")
#define ORC_PAR_SUM(v) ORC_PRAGMA(omp parallel for reduction(+ : v) schedule(static) if (n >= ORC_OMP_MIN))
#define ORC_PAR_MAX(v) ORC_PRAGMA(omp parallel for reduction(max : v) schedule(static) if (n >= ORC_OMP_MIN))
static void par_copy(double *d, const double *s, int64_t n)
{
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) d[i] = s[i];
}
#else
#define ORC_PAR
#define ORC_PAR_SUM(v)
#define ORC_PAR_MAX(v)
static void par_copy(double *d, const double *s, int64_t n) { memcpy(d, s, sizeof(double) * (size_t)n); }
#endif

/* ------------------------------------------------------------------ */
/* BLAS-1 substrate (L0 of SURVEY.md §1)                               */
/* ------------------------------------------------------------------ */

#ifdef ORC_EXACT_SUMS
/* ---- twice-working-precision reductions (the arbiter build) -------------------------------------------------- */
#define ORC_BLK 65536
typedef struct { double hi, lo; } dd_t;
static inline void two_sum(double a, double b, double *s, double *e)
{
    const double t = a + b, z = t - a;
    *s = t;
    *e = (a - (t - z)) + (b - z);
}
static inline dd_t dd_add(dd_t acc, double hi, double lo)
{
    double e;
    two_sum(acc.hi, hi, &acc.hi, &e);
    acc.lo += e + lo;
    return acc;
}
/* one block: four independent (p, s) chains (the TwoSum chain is latency-bound), folded in double-double */
static dd_t dd_block(const double *a, const double *b, int64_t n, int is_sum)
{
    double p[4] = {0, 0, 0, 0}, s[4] = {0, 0, 0, 0};
    int64_t i = 0;
    for (; i + 4 <= n; i += 4)
        for (int k = 0; k < 4; ++k) {
            double h, r = 0.0, q;
            if (is_sum) h = a[i + k];
            else { h = a[i + k] * b[i + k]; r = __builtin_fma(a[i + k], b[i + k], -h); } /* TwoProduct */
            two_sum(p[k], h, &p[k], &q);
            s[k] += q + r;
        }
    for (; i < n; ++i) {
        double h, r = 0.0, q;
        if (is_sum) h = a[i];
        else { h = a[i] * b[i]; r = __builtin_fma(a[i], b[i], -h); }
        two_sum(p[0], h, &p[0], &q);
        s[0] += q + r;
    }
    dd_t acc = {0.0, 0.0};
    for (int k = 0; k < 4; ++k) acc = dd_add(acc, p[k], s[k]);
    return acc;
}
static double dd_reduce(const double *a, const double *b, int64_t n, int is_sum)
{
    const int64_t nblk = (n + ORC_BLK - 1) / ORC_BLK;
    if (nblk <= 1) { const dd_t r = dd_block(a, b, n, is_sum); return r.hi + r.lo; }
    dd_t *blk = (dd_t *)malloc(sizeof(dd_t) * (size_t)nblk);
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nblk; ++k) {
        const int64_t lo = k * ORC_BLK, len = (lo + ORC_BLK <= n) ? ORC_BLK : n - lo;
        blk[k] = dd_block(a + lo, is_sum ? NULL : b + lo, len, is_sum);
    }
    dd_t acc = {0.0, 0.0};
    for (int64_t k = 0; k < nblk; ++k) acc = dd_add(acc, blk[k].hi, blk[k].lo); /* fixed order: thread-count independent */
    free(blk);
    return acc.hi + acc.lo;
}
/* Σ v_i of already rounded terms, in twice the working precision (the objective sums) */
double orc_sum(const double *v, int64_t n) { return dd_reduce(v, NULL, n, 1); }
/* the pin of Dot2: the same dot product accumulated in __float128 (113-bit significand: every product of two doubles
 * is exact, the sum is rounded 2^-113 per addition), rounded to double once */
double orc_dot_f128(const double *a, const double *b, int64_t n)
{
    __float128 t = 0;
    for (int64_t i = 0; i < n; ++i) t += (__float128)a[i] * (__float128)b[i];
    return (double)t;
}
double orc_sum_f128(const double *v, int64_t n)
{
    __float128 t = 0;
    for (int64_t i = 0; i < n; ++i) t += (__float128)v[i];
    return (double)t;
}
int orc_exact_sums(void) { return 1; }
#else
int orc_exact_sums(void) { return 0; }
#endif

/* LinearAlgebra.dot → BLAS ddot.  8 independent partial sums, the shape of
 * an unrolled SIMD ddot micro-kernel; order is unspecified in the reference. */
double orc_dot(const double *a, const double *b, int64_t n)
{
#ifdef ORC_EXACT_SUMS
    return dd_reduce(a, b, n, 0);
#endif
#ifdef _OPENMP
    if (n >= ORC_OMP_MIN) { /* all-cores baseline build only (libcgo_oracle_omp.so) */
        double t = 0.0;
#pragma omp parallel for reduction(+ : t) schedule(static)
        for (int64_t i = 0; i < n; ++i) t += a[i] * b[i];
        return t;
    }
#endif
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        s0 += a[i] * b[i];
        s1 += a[i + 1] * b[i + 1];
        s2 += a[i + 2] * b[i + 2];
        s3 += a[i + 3] * b[i + 3];
        s4 += a[i + 4] * b[i + 4];
        s5 += a[i + 5] * b[i + 5];
        s6 += a[i + 6] * b[i + 6];
        s7 += a[i + 7] * b[i + 7];
    }
    double s = ((s0 + s4) + (s2 + s6)) + ((s1 + s5) + (s3 + s7));
    for (; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* LinearAlgebra.norm (2-norm) = BLAS.nrm2 (n ≥ 32) / generic_norm2: the true 2-norm, finite
 * whenever it is representable.  Fast path sqrt(Σ a_i²) while the squares neither overflow nor
 * underflow; otherwise the scaled form maxabs·sqrt(Σ (a_i/maxabs)²) of generic_norm2. */
double orc_norm(const double *a, int64_t n)
{
    const double ss = orc_dot(a, a, n);
    if (ss >= 1e-280 && ss <= 1e300) return sqrt(ss);
    double maxabs = 0.0;
    int has_nan = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double v = fabs(a[i]);
        if (isnan(v)) has_nan = 1;
        if (v > maxabs) maxabs = v;
    }
    if (has_nan) return NAN;
    if (maxabs == 0.0 || isinf(maxabs)) return maxabs;
    double t = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double r = a[i] / maxabs;
        t += r * r;
    }
    return maxabs * sqrt(t);
}

static double jl_max(double a, double b) /* Base.max: NaN-propagating */
{
    if (isnan(a) || isnan(b)) return NAN;
    return a > b ? a : b;
}
static double jl_min(double a, double b) /* Base.min: NaN-propagating */
{
    if (isnan(a) || isnan(b)) return NAN;
    return a < b ? a : b;
}

static const char *k_status_names[ORC_NUM_STATUS] = {
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

const char *orc_status_name(int s)
{
    if (s < 0 || s >= ORC_NUM_STATUS) return "unknown";
    return k_status_names[s];
}

/* ------------------------------------------------------------------ */
/* config checks = the reference's @assert sites                       */
/* ------------------------------------------------------------------ */
int orc_check_cg_config(const orc_cg_config *c)
{
    if (!(0.0 < c->eps && c->eps < 1.0)) return 187; /* types.jl:187 */
    if (c->beta.kind < 0 || c->beta.kind > ORC_BETA_BROYDEN_FAMILY) return 1;
    if (c->beta.kind == ORC_BETA_BROYDEN_FAMILY && !(0.0 <= c->beta.mu)) return 57; /* qn_flavours.jl:57 */
    if (c->beta.kind == ORC_BETA_LBFGS && c->beta.lbfgs_m < 1) return 2;
    return 0;
}

int orc_check_ls_config(const orc_ls_config *l)
{
    if (l->kind == ORC_LS_STRONG_WOLFE_BISECTION) {
        if (!(0.0 < l->c1 && l->c1 < l->c2 && l->c2 < 1.0)) return 22; /* nocedal.jl:22 */
        if (!(l->max_iters >= 0)) return 24;                           /* nocedal.jl:24 */
        if (!(l->zoom_max_iters >= 0)) return 25;                      /* nocedal.jl:25 */
        if (!(l->a_max_growth_factor > 1.0)) return 26;                /* nocedal.jl:26 */
        return 0;
    }
    if (l->kind == ORC_LS_WOLFE_BISECTION) {
        if (l->cond_kind == ORC_COND_WOLFE) {
            if (!(0.0 < l->c1 && l->c1 < l->c2 && l->c2 < 1.0)) return 278; /* wolfe.jl:278 */
        } else if (l->cond_kind == ORC_COND_YUAN_WEI_LU) {
            if (!(0.0 < l->delta1 && l->delta1 < l->c1 && l->c1 < l->c2 && l->c2 < 1.0))
                return 233; /* wolfe.jl:233 */
        } else
            return 3;
        return 0;
    }
    if (l->kind == ORC_LS_BACKTRACKING) {
        if (l->cond_kind != ORC_COND_ARMIJO) return 3;
        if (!(0.0 < l->c1 && l->c1 < 1.0)) return 169; /* geometric.jl:169 */
        return 0;
    }
    return 4;
}

/* ------------------------------------------------------------------ */
/* work state                                                          */
/* ------------------------------------------------------------------ */
typedef struct { /* LineSearchContainer, types.jl:84-100 */
    double *xp, *df_xp, *x, *u;
} ls_container;

typedef struct { /* L-BFGS ring (NEW QNβConfig state; contract qn_flavours.jl:46-48) */
    int m, count, head; /* head = slot of the newest pair */
    double *S, *Y;      /* m × n each */
    double *rho, *alpha;
    double gamma;
    double *q;
    double *ts, *ty; /* candidate pair, committed to the ring only if s·y > 0 */
} lbfgs_state;

typedef struct {
    orc_fdf_t fdf;
    void *user;
    int64_t n;
    orc_results *ret;
    /* getβ temporaries (cg_flavours.jl allocates these per call) */
    double *y, *tmp1, *tmp2;
    lbfgs_state qn;
    int64_t total_evals;
} solver;

static void log_eval(solver *S, double a, double phi, double dphi)
{
    orc_results *r = S->ret;
    if (r && r->log_a && r->log_len < r->log_cap) {
        r->log_a[r->log_len] = a;
        r->log_phi[r->log_len] = phi;
        r->log_dphi[r->log_len] = dphi;
        if (r->log_margin) r->log_margin[r->log_len] = INFINITY;
    }
    if (r) r->log_len++;
}

/* Decision margin of a branch taken on the LAST logged evaluation: |lhs − rhs| / scale (scale = the larger magnitude
 * of the two sides unless the caller knows the natural one), the smallest over that evaluation's branches.  A branch
 * whose margin is ≳ the perturbation a different summation order can cause is taken alike by every implementation;
 * the arbiter build's log says where that stops being true. */
static void note_margin(solver *S, double lhs, double rhs, double scale)
{
    orc_results *r = S->ret;
    if (!r || !r->log_margin || r->log_len < 1 || r->log_len > r->log_cap) return;
    if (!(scale > 0.0)) scale = fmax(fabs(lhs), fabs(rhs));
    double m = fabs(lhs - rhs) / scale;
    if (!(scale > 0.0) || isnan(m)) m = 0.0;
    if (m < r->log_margin[r->log_len - 1]) r->log_margin[r->log_len - 1] = m;
}

/* cg_utils.jl:4-23  evalϕdϕ! */
static void eval_phi_dphi(solver *S, ls_container *info, double a, double *phi, double *dphi)
{
    const int64_t n = S->n;
    const double *x = info->x, *u = info->u;
    double *xp = info->xp;
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) xp[i] = x[i] + a * u[i]; /* :14-16 */
    *phi = S->fdf(S->user, info->df_xp, xp, n);              /* :19 */
    S->total_evals++;
    *dphi = orc_dot(info->df_xp, u, n);                      /* :20 */
    log_eval(S, a, *phi, *dphi);
}

/* cg_flavours.jl:2-15  updatedir! (CG) */
void orc_updatedir(double *u, const double *df_x, double beta, int64_t n)
{
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) u[i] = -df_x[i] + beta * u[i]; /* :10-12 */
}

/* ------------------------------------------------------------------ */
/* getβ                                                                */
/* ------------------------------------------------------------------ */
static double getbeta_impl(const orc_beta_config *b, const double *gn, const double *g,
                           const double *u, int64_t n, double *y, double *tmp1, double *tmp2)
{
    switch (b->kind) {
    case ORC_BETA_YUAN_WANG_SHENG: { /* cg_flavours.jl:51-79 */
        const double mu = b->mu;
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) y[i] = gn[i] - g[i];              /* :63 */
        const double R1 = mu * orc_norm(u, n) * orc_norm(y, n);          /* :65 */
        const double R2 = orc_dot(u, y, n);                              /* :66 */
        const double R3 = 2 * orc_dot(y, y, n) * orc_dot(u, gn, n) / orc_dot(y, gn, n); /* :67 */
        const double R = jl_max(jl_max(R1, R2), R3);                     /* :68 */
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) tmp2[i] = gn[i] / R;             /* :71 */
        const double m = 2 * orc_dot(y, y, n) / R;                       /* :73 */
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) tmp1[i] = y[i] - m * u[i];       /* :74 */
        return orc_dot(tmp1, tmp2, n);                                   /* :76 */
    }
    case ORC_BETA_HAGER_ZHANG: { /* cg_flavours.jl:87-108 */
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) y[i] = gn[i] - g[i];             /* :96 */
        const double R = orc_dot(u, y, n);                               /* :98 */
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) tmp2[i] = gn[i] / R;             /* :100 */
        const double m = 2 * orc_dot(y, y, n) / R;                       /* :102 */
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) tmp1[i] = y[i] - m * u[i];       /* :103 */
        return orc_dot(tmp1, tmp2, n);                                   /* :105 */
    }
    case ORC_BETA_SALLEH_ALHAWARAT: { /* cg_flavours.jl:133-151 */
        const double nrm = orc_norm(gn, n);
        const double norm_sq = nrm * nrm;                                /* :140 norm(g_next)^2 */
        const double tmp = orc_dot(gn, g, n);                            /* :141 */
        if (norm_sq > tmp) {                                             /* :143 */
            const double numerator = norm_sq - tmp;
            const double denominator = orc_dot(u, gn, n) - orc_dot(u, g, n); /* :145 */
            return numerator / denominator;
        }
        return 0.0;                                                      /* :150 */
    }
    case ORC_BETA_LIU_STORREY: { /* cg_flavours.jl:157-170 */
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) y[i] = gn[i] - g[i];             /* :164 */
        const double numerator = orc_dot(gn, y, n);                      /* :166 */
        const double denominator = -orc_dot(u, y, n);                    /* :167 */
        return numerator / denominator;
    }
    case ORC_BETA_HESTENES_STIEFEL: { /* NEW; the commented body at cg_flavours.jl:121-126 */
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) y[i] = gn[i] - g[i];
        return orc_dot(gn, y, n) / orc_dot(u, y, n);
    }
    case ORC_BETA_POLAK_RIBIERE: { /* NEW: β = g⁺·(g⁺−g) / g·g */
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) y[i] = gn[i] - g[i];
        return orc_dot(gn, y, n) / orc_dot(g, g, n);
    }
    case ORC_BETA_DAI_YUAN: { /* NEW: β = g⁺·g⁺ / u·(g⁺−g) */
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) y[i] = gn[i] - g[i];
        return orc_dot(gn, gn, n) / orc_dot(u, y, n);
    }
    default:
        return NAN;
    }
}

double orc_getbeta(const orc_beta_config *b, const double *gn, const double *g, const double *u,
                   int64_t n)
{
    double *w = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
    const double r = getbeta_impl(b, gn, g, u, n, w, w + n, w + 2 * n);
    free(w);
    return r;
}

/* ------------------------------------------------------------------ */
/* L-BFGS (NEW; behind the QN dispatch contract of qn_flavours.jl:5-48: */
/* getβ mutates+returns the state, updatedir!(u, g, state) sets u=−Hg)  */
/* ------------------------------------------------------------------ */
static void lbfgs_init(lbfgs_state *q, int m, int64_t n)
{
    memset(q, 0, sizeof(*q));
    q->m = m;
    q->head = -1;
    q->gamma = 1.0;
    q->S = (double *)malloc(sizeof(double) * (size_t)m * (size_t)n);
    q->Y = (double *)malloc(sizeof(double) * (size_t)m * (size_t)n);
    q->rho = (double *)calloc((size_t)m, sizeof(double));
    q->alpha = (double *)calloc((size_t)m, sizeof(double));
    q->q = (double *)malloc(sizeof(double) * (size_t)n);
    q->ts = (double *)malloc(sizeof(double) * (size_t)n);
    q->ty = (double *)malloc(sizeof(double) * (size_t)n);
}
static void lbfgs_free(lbfgs_state *q)
{
    free(q->S); free(q->Y); free(q->rho); free(q->alpha); free(q->q); free(q->ts); free(q->ty);
    memset(q, 0, sizeof(*q));
}

/* "getβ" for L-BFGS: push (s = a*·u, y = g⁺ − g).  The pair is skipped when
 * s·y is not strictly positive (curvature condition failed / NaN). */
static void lbfgs_push(lbfgs_state *q, const double *gn, const double *g, const double *u,
                       double a_star, int64_t n)
{
    const int slot = (q->head + 1) % q->m;
    double *s = q->ts, *y = q->ty;
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) {
        s[i] = a_star * u[i];
        y[i] = gn[i] - g[i];
    }
    const double sy = orc_dot(s, y, n);
    const double yy = orc_dot(y, y, n);
    if (!(sy > 0.0)) return; /* pair dropped; the ring (incl. its oldest pair when full) is untouched */
    par_copy(q->S + (size_t)slot * (size_t)n, s, n);
    par_copy(q->Y + (size_t)slot * (size_t)n, y, n);
    q->rho[slot] = 1.0 / sy;
    q->gamma = sy / yy;
    q->head = slot;
    if (q->count < q->m) q->count++;
}

/* "updatedir!" for L-BFGS: two-loop recursion (Nocedal & Wright Alg. 7.4), u = −H·g */
static void lbfgs_updatedir(lbfgs_state *q, double *u, const double *df_x, int64_t n)
{
    double *r = q->q;
    par_copy(r, df_x, n);
    for (int k = 0; k < q->count; ++k) { /* newest → oldest */
        const int slot = ((q->head - k) % q->m + q->m) % q->m;
        const double *s = q->S + (size_t)slot * (size_t)n, *y = q->Y + (size_t)slot * (size_t)n;
        const double al = q->rho[slot] * orc_dot(s, r, n);
        q->alpha[slot] = al;
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) r[i] = r[i] - al * y[i];
    }
    const double gam = q->count > 0 ? q->gamma : 1.0;
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) r[i] = gam * r[i];
    for (int k = q->count - 1; k >= 0; --k) { /* oldest → newest */
        const int slot = ((q->head - k) % q->m + q->m) % q->m;
        const double *s = q->S + (size_t)slot * (size_t)n, *y = q->Y + (size_t)slot * (size_t)n;
        const double b = q->rho[slot] * orc_dot(y, r, n);
        const double c = q->alpha[slot] - b;
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) r[i] = r[i] + c * s[i];
    }
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) u[i] = -r[i];
}

/* ------------------------------------------------------------------ */
/* Wolfe conditions  (wolfe.jl:213-294)                                */
/* ------------------------------------------------------------------ */
/* ------------------------------------------------------------------ */
/* qn_flavours.jl:3-90  BroydenFamily — dense, as written (small n only) */
/* ------------------------------------------------------------------ */
static int dense_isposdef(const double *B, int64_t n) /* LinearAlgebra.isposdef: Hermitian + Cholesky succeeds */
{
    for (int64_t i = 0; i < n * n; ++i) if (!isfinite(B[i])) return 0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < i; ++j) if (B[i * n + j] != B[j * n + i]) return 0;
    double *L = (double *)malloc(sizeof(double) * (size_t)(n * n));
    memcpy(L, B, sizeof(double) * (size_t)(n * n));
    int ok = 1;
    for (int64_t j = 0; j < n && ok; ++j) {
        double d = L[j * n + j];
        for (int64_t k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        if (!(d > 0.0)) { ok = 0; break; }
        d = sqrt(d);
        L[j * n + j] = d;
        for (int64_t i = j + 1; i < n; ++i) {
            double t = L[i * n + j];
            for (int64_t k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = t / d;
        }
    }
    free(L);
    return ok;
}

static void dense_solve(const double *B, const double *rhs, double *out, int64_t n) /* B\rhs: LU, partial pivoting */
{
    double *A = (double *)malloc(sizeof(double) * (size_t)(n * n));
    memcpy(A, B, sizeof(double) * (size_t)(n * n));
    memcpy(out, rhs, sizeof(double) * (size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        int64_t p = k;
        for (int64_t i = k + 1; i < n; ++i) if (fabs(A[i * n + k]) > fabs(A[p * n + k])) p = i;
        if (p != k) {
            for (int64_t j = 0; j < n; ++j) { const double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            const double t = out[k]; out[k] = out[p]; out[p] = t;
        }
        for (int64_t i = k + 1; i < n; ++i) {
            const double m = A[i * n + k] / A[k * n + k];
            if (m != 0.0) {
                for (int64_t j = k + 1; j < n; ++j) A[i * n + j] -= m * A[k * n + j];
                out[i] -= m * out[k];
            }
        }
    }
    for (int64_t i = n - 1; i >= 0; --i) {
        double t = out[i];
        for (int64_t j = i + 1; j < n; ++j) t -= A[i * n + j] * out[j];
        out[i] = t / A[i * n + i];
    }
    free(A);
}

static void dense_identity(double *B, int64_t n)
{
    memset(B, 0, sizeof(double) * (size_t)(n * n));
    for (int64_t i = 0; i < n; ++i) B[i * n + i] = 1.0;
}

/* updatedir!(u, df_x, B)  qn_flavours.jl:3-22 (also initializeLineSearchContainer! :25-44) */
static void broyden_updatedir(double *B, double *u, const double *df_x, int64_t n, double *tmp)
{
    if (!dense_isposdef(B, n)) dense_identity(B, n);
    for (int64_t i = 0; i < n; ++i) tmp[i] = -df_x[i];
    dense_solve(B, tmp, u, n);
}

/* getβ(::BroydenFamily, g_next, g, u)  qn_flavours.jl:70-90 */
static void broyden_getbeta(double theta, double *B, const double *gn, const double *g, int64_t n)
{
    double *y = (double *)malloc(sizeof(double) * (size_t)n), *s = (double *)malloc(sizeof(double) * (size_t)n);
    double *Bs = (double *)malloc(sizeof(double) * (size_t)n), *v = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) y[i] = gn[i] - g[i];     /* :78 */
    dense_solve(B, y, s, n);                                 /* :81  s = B\y */
    for (int64_t i = 0; i < n; ++i) {                        /* :83  Bs = B*s */
        double t = 0.0;
        for (int64_t j = 0; j < n; ++j) t += B[i * n + j] * s[j];
        Bs[i] = t;
    }
    const double sBs = orc_dot(s, Bs, n), sy = orc_dot(s, y, n);
    const double tmp = theta * sBs;
    for (int64_t i = 0; i < n; ++i) v[i] = y[i] / sy - Bs[i] / sBs;   /* :86 */
    for (int64_t i = 0; i < n; ++i)                                      /* :87 */
        for (int64_t j = 0; j < n; ++j)
            B[i * n + j] = B[i * n + j] - Bs[i] * Bs[j] / sBs + y[i] * y[j] / sy + tmp * v[i] * v[j];
    free(y); free(s); free(Bs); free(v);
}

void orc_evalwolfeconditions(const orc_ls_config *ls, double phi_a, double dphi_a, double a,
                             const double *u, int64_t n, double phi_0, double dphi_0,
                             int *valid_large, int *valid_small)
{
    const double c1 = ls->c1, c2 = ls->c2;
    if (ls->cond_kind == ORC_COND_YUAN_WEI_LU) { /* wolfe.jl:219-251 */
        const double d1 = ls->delta1;
        const double norm_u_sq = orc_dot(u, u, n);                               /* :240 */
        const double RHS1 = phi_0 + c1 * a * dphi_0 +
                            a * jl_min(-d1 * dphi_0, c1 * a * norm_u_sq / 2);    /* :243 */
        *valid_large = phi_a <= RHS1;                                            /* :244 */
        const double RHS2 = c2 * dphi_0 + jl_min(-d1 * dphi_0, c1 * a * norm_u_sq); /* :247 */
        *valid_small = dphi_a >= RHS2;                                           /* :248 */
    } else { /* wolfe.jl:264-294 */
        const double RHS1 = phi_0 + c1 * a * dphi_0; /* :286 */
        *valid_large = phi_a <= RHS1;                /* :287 */
        const double RHS2 = c2 * dphi_0;             /* :290 */
        *valid_small = dphi_a >= RHS2;               /* :291 */
    }
}

/* ------------------------------------------------------------------ */
/* nocedal.jl:162-209  zoom!                                           */
/* ------------------------------------------------------------------ */
static void zoom(solver *S, ls_container *info, double a_lb, double a_ub, double phi_a_lb,
                 double phi_0, double dphi_0, double c1, double c2, int64_t evals,
                 int64_t max_iters, double *o_phi, double *o_a, int64_t *o_evals, int *o_status)
{
    double a = 0, phi_a = 0, dphi_a = 0; /* :179-181 */
    for (int64_t it = 0; it < max_iters; ++it) {
        a = (a_lb + a_ub) / 2;                                  /* :186 */
        eval_phi_dphi(S, info, a, &phi_a, &dphi_a);             /* :189 */
        evals += 1;
        note_margin(S, phi_a, phi_0 + c1 * a * dphi_0, 0.0);
        if (!(phi_a > phi_0 + c1 * a * dphi_0)) note_margin(S, phi_a, phi_a_lb, 0.0);
        if ((phi_a > phi_0 + c1 * a * dphi_0) || (phi_a >= phi_a_lb)) { /* :192 */
            a_ub = a;
        } else {
            note_margin(S, fabs(dphi_a), -c2 * dphi_0, 0.0);
            if (!(fabs(dphi_a) <= -c2 * dphi_0)) note_margin(S, dphi_a, 0.0, fabs(dphi_0));
            if (fabs(dphi_a) <= -c2 * dphi_0) {                 /* :195 */
                *o_phi = phi_a; *o_a = a; *o_evals = evals; *o_status = ORC_SUCCESS;
                return;
            }
            if (dphi_a * (a_ub - a_lb) >= 0) a_ub = a_lb;       /* :199-201 */
            a_lb = a;                                           /* :202 */
            phi_a_lb = phi_a;                                   /* :203 */
        }
    }
    *o_phi = phi_a; *o_a = a; *o_evals = evals; *o_status = ORC_ZOOM_MAX_ITERS_REACHED; /* :208 */
}

/* nocedal.jl:33-158  linesearch!(::StrongWolfeBisection) */
static void linesearch_strong_wolfe(solver *S, ls_container *info, const orc_ls_config *cfg,
                                    double f_x, const double *df_x, double a_initial,
                                    double *o_phi, double *o_a, int64_t *o_evals, int *o_status)
{
    const int64_t max_iters = cfg->max_iters, zoom_max_iters = cfg->zoom_max_iters;
    const double c1 = cfg->c1, c2 = cfg->c2, growth = cfg->a_max_growth_factor;

    if (!(0.0 < a_initial && isfinite(a_initial))) a_initial = 1.0; /* :49-52 */

    const double phi_0 = f_x;                                       /* :55 */
    const double dphi_0 = orc_dot(df_x, info->u, S->n);             /* :56 */
    if (dphi_0 > 0.0) {                                             /* :57-63 */
        *o_phi = phi_0; *o_a = 0.0; *o_evals = 0; *o_status = ORC_NON_DESCENT_SEARCH_DIRECTION;
        return;
    }
    double a_prev = 0.0, phi_a_prev = phi_0;                        /* :65-66 */
    double a = a_initial, phi_a = phi_0, dphi_a = dphi_0;           /* :68-70 */
    double a_max = a * growth;                                      /* :71 */
    int64_t evals = 0;
    int non_initial_iter = 0;
    for (int64_t it = 0; it < max_iters; ++it) {                    /* :76 */
        eval_phi_dphi(S, info, a, &phi_a, &dphi_a);                 /* :78 */
        evals += 1;
        const int chk1 = phi_a > phi_0 + c1 * a * dphi_0;           /* :81 */
        const int chk2 = phi_a >= phi_a_prev;                       /* :82 */
        note_margin(S, phi_a, phi_0 + c1 * a * dphi_0, 0.0);
        if (!chk1 && non_initial_iter) note_margin(S, phi_a, phi_a_prev, 0.0);
        if (!(chk1 || (chk2 && non_initial_iter))) {
            note_margin(S, fabs(dphi_a), -c2 * dphi_0, 0.0);
            if (!(fabs(dphi_a) <= -c2 * dphi_0)) note_margin(S, dphi_a, 0.0, fabs(dphi_0));
        }
        if (chk1 || (chk2 && non_initial_iter)) {                   /* :83-105 */
            zoom(S, info, a_prev, a, phi_a_prev, phi_0, dphi_0, c1, c2, evals, zoom_max_iters,
                 o_phi, o_a, o_evals, o_status);
            return;
        }
        if (fabs(dphi_a) <= -c2 * dphi_0) {                         /* :107-110 */
            *o_phi = phi_a; *o_a = a; *o_evals = evals; *o_status = ORC_SUCCESS;
            return;
        }
        if (dphi_a >= 0) {                                          /* :112-134 */
            zoom(S, info, a, a_prev, phi_a, phi_0, dphi_0, c1, c2, evals, zoom_max_iters, o_phi,
                 o_a, o_evals, o_status);
            return;
        }
        a_prev = a;                                                 /* :137 */
        phi_a_prev = phi_a;                                         /* :138 */
        non_initial_iter = 1;                                       /* :139 */
        a_max = a * growth;                                         /* :143 */
        if (a > a_max) {                                            /* :144-149 */
            *o_phi = phi_a; *o_a = a; *o_evals = evals; *o_status = ORC_LINESEARCH_A_MAX_OVERFLOW;
            return;
        }
        a = (a_max + a) / 2;                                        /* :150 */
    }
    *o_phi = phi_a; *o_a = a; *o_evals = evals; *o_status = ORC_LINESEARCH_MAX_ITERS_REACHED; /* :157 */
}

/* wolfe.jl:171-207  findfeasiblestepsize! */
static int findfeasiblestepsize(solver *S, ls_container *info, int64_t *evals, double *a_io,
                                double reduction_factor, double lb, int64_t max_iters,
                                double *phi_a, double *dphi_a)
{
    double a = *a_io;
    if (lb > a) { /* :186-188 */
        *phi_a = 0.0; *dphi_a = 0.0;
        return ORC_BISECTION_LOWER_BOUND_LARGER_THAN_PROPOSED_STEP;
    }
    eval_phi_dphi(S, info, a, phi_a, dphi_a); /* :191 */
    *evals += 1;
    int64_t iter = 1;
    while (a > lb && iter < max_iters) {      /* :195 */
        if (isfinite(*phi_a) && isfinite(*dphi_a)) { /* :196-198 */
            *a_io = a;
            return ORC_FEASIBLE;
        }
        a = a * reduction_factor;             /* :200 */
        eval_phi_dphi(S, info, a, phi_a, dphi_a);
        *evals += 1;
        iter += 1;
    }
    *a_io = a;
    return ORC_INFEASIBLE;                    /* :206 */
}

/* wolfe.jl:13-165  linesearch!(::WolfeBisection) */
static void linesearch_wolfe_bisection(solver *S, ls_container *info, const orc_ls_config *cfg,
                                       double f_x, double *df_x, double a_initial, double *o_phi,
                                       double *o_a, int64_t *o_evals, int *o_status)
{
    const int64_t n = S->n;
    const double reduction_factor = 0.5, growth_factor = 2.0;        /* :23-24 */
    const double max_step_size = cfg->max_step_size;
    const int64_t max_iters = cfg->max_iters, feas_iters = cfg->feasibility_max_iters;
    double *u = info->u;

    if (!(max_step_size > a_initial && a_initial > 0.0))             /* :30-32 */
        a_initial = jl_min(1.0, max_step_size / 2);

    double phi_0 = f_x;                                              /* :35 */
    if (!isfinite(phi_0)) {                                          /* :36-38 */
        *o_phi = phi_0; *o_a = 0.0; *o_evals = 0; *o_status = ORC_ACCEPTED_NON_FINITE_ITERATE;
        return;
    }
    const double dphi_0 = orc_dot(df_x, u, n);                       /* :40 */
    if (dphi_0 > 0.0) {                                              /* :41-43 */
        *o_phi = phi_0; *o_a = 0.0; *o_evals = 0; *o_status = ORC_NON_DESCENT_SEARCH_DIRECTION;
        return;
    }
    double a = a_initial;
    int64_t evals = 0;
    double lb = 0.0, ub = INFINITY;                                  /* :47-48 */
    double phi_a, dphi_a;
    int flag = findfeasiblestepsize(S, info, &evals, &a, reduction_factor, 0.0, feas_iters,
                                    &phi_a, &dphi_a);                /* :51-62 */
    if (flag != ORC_FEASIBLE) {                                      /* :63-65 */
        *o_phi = phi_0; *o_a = 0.0; *o_evals = 0; *o_status = ORC_CANNOT_FIND_INITIAL_FEASIBLE_STEP;
        return;
    }
    for (int64_t it = 0; it < max_iters; ++it) {                     /* :67 */
        int valid_large, valid_small;
        orc_evalwolfeconditions(cfg, phi_a, dphi_a, a, u, n, phi_0, dphi_0, &valid_large,
                                &valid_small);                       /* :70-78 */
        if (S->ret && S->ret->log_margin) {   /* the two inequalities again, for their margins (same arithmetic) */
            double rhs1, rhs2;
            if (cfg->cond_kind == ORC_COND_YUAN_WEI_LU) {
                const double nu = orc_dot(u, u, n);
                rhs1 = phi_0 + cfg->c1 * a * dphi_0 + a * jl_min(-cfg->delta1 * dphi_0, cfg->c1 * a * nu / 2);
                rhs2 = cfg->c2 * dphi_0 + jl_min(-cfg->delta1 * dphi_0, cfg->c1 * a * nu);
            } else {
                rhs1 = phi_0 + cfg->c1 * a * dphi_0;
                rhs2 = cfg->c2 * dphi_0;
            }
            note_margin(S, phi_a, rhs1, 0.0);
            if (valid_large) note_margin(S, dphi_a, rhs2, fmax(fabs(dphi_a), fabs(dphi_0)));
        }
        if (!valid_large || !valid_small) {                          /* :80 */
            if (!valid_large) {
                ub = a;                                              /* :86 */
                a = (lb + ub) / 2;                                   /* :95 */
            } else {
                lb = a;                                              /* :98 */
                if (!isfinite(ub)) {                                 /* :99 */
                    a = growth_factor * a;                           /* :102 */
                    if (a > max_step_size) {                         /* :104-112 */
                        *o_phi = phi_0; *o_a = 0.0; *o_evals = 0;
                        *o_status = ORC_MAX_STEP_LENGTH_REACHED;
                        return;
                    }
                } else {
                    a = (lb + ub) / 2;                               /* :114 */
                }
            }
            if (!(lb < a && a < ub)) {                               /* :122 */
                /* norm(u + df_x), with the u+df_x temporary (wolfe.jl:123) */
                double *t = S->tmp1;
                ORC_PAR
                for (int64_t i = 0; i < n; ++i) t[i] = u[i] + df_x[i];
                const double nr = orc_norm(t, n);
                /* isapprox(nr, 0) with default rtol/atol ≡ (nr == 0) */
                if (!(nr == 0.0)) {
                    lb = 0.0;                                        /* :125 */
                    ub = INFINITY;                                   /* :126 */
                    a = a_initial;                                   /* :128 */
                    for (int64_t i = 0; i < n; ++i) u[i] = -df_x[i]; /* :129 */
                } else {
                    /* wolfe.jl:131 — bare tuple expression, no `return`: a no-op */
                }
            }
            flag = findfeasiblestepsize(S, info, &evals, &a, reduction_factor, lb, feas_iters,
                                        &phi_a, &dphi_a);            /* :141-152 */
            if (flag != ORC_FEASIBLE) {                              /* :153-158 */
                *o_phi = phi_0; *o_a = 0.0; *o_evals = 0;
                *o_status = ORC_CANNOT_FIND_FEASIBLE_STEP;
                return;
            }
        } else {
            *o_phi = phi_a; *o_a = a; *o_evals = evals; *o_status = ORC_SUCCESS; /* :160 */
            return;
        }
    }
    *o_phi = phi_a; *o_a = a; *o_evals = evals; *o_status = ORC_LINESEARCH_MAX_ITERS_REACHED; /* :164 */
}

/* ------------------------------------------------------------------ */
/* geometric.jl:15-186  Backtracking / Armijo — restated bug for bug:   */
/*  - one redundant evaluation at geometric.jl:78;                      */
/*  - on :success the PREVIOUS (ϕ, a) is returned while info.xp /       */
/*    info.df_xp hold the last (rejected) trial (geometric.jl:141-144), */
/*    which the outer loop then adopts (optim.jl:136-139);              */
/*  - the shrink branch returns an Armijo-violating step as :success.   */
/* ------------------------------------------------------------------ */
static int armijo_ok(double c1, double phi_a, double a, double phi_0, double dphi_0)
{                                                                    /* geometric.jl:164-186 */
    if (!isfinite(phi_0) || !isfinite(phi_a) || !isfinite(a)) return 0; /* :175-177 */
    const double LHS1 = phi_0 - phi_a;                                /* :180 */
    return LHS1 >= -c1 * a * dphi_0;                                  /* :181 */
}

static void geometricsearch(solver *S, ls_container *info, const orc_ls_config *cfg, double a,
                            int divide, int64_t evals, double phi_a, double phi_0, double dphi_0,
                            double *o_phi, double *o_a, int64_t *o_evals, int *o_status)
{                                                                    /* geometric.jl:102-152 */
    const double rho = cfg->discount_factor;
    double a_prev = a, phi_a_prev = phi_a, dphi_a;
    for (int64_t it = 0; it < cfg->max_iters; ++it) {
        a = divide ? a / rho : a * rho;                               /* :126 (getgeometricstep :7-13) */
        if (!isfinite(a)) {                                           /* :127-129 */
            *o_phi = phi_a_prev; *o_a = a_prev; *o_evals = evals; *o_status = ORC_NON_FINITE_STEP_PROPOSED;
            return;
        }
        if (a == a_prev) {                                            /* :131-133 */
            *o_phi = phi_a_prev; *o_a = a_prev; *o_evals = evals;
            *o_status = ORC_PROPOSED_STEP_SAME_AS_CURRENT_STEP;
            return;
        }
        eval_phi_dphi(S, info, a, &phi_a, &dphi_a);                   /* :136 */
        evals += 1;
        note_margin(S, phi_0 - phi_a, -cfg->c1 * a * dphi_0, fmax(fabs(phi_0), fabs(phi_a)));
        if (!armijo_ok(cfg->c1, phi_a, a, phi_0, dphi_0)) {           /* :139-143 */
            *o_phi = phi_a_prev; *o_a = a_prev; *o_evals = evals; *o_status = ORC_SUCCESS;
            return;
        }
        a_prev = a;                                                   /* :146 */
        phi_a_prev = phi_a;                                           /* :147 */
    }
    *o_phi = phi_a; *o_a = a; *o_evals = evals; *o_status = ORC_LINESEARCH_MAX_ITERS_REACHED; /* :150 */
}

static void linesearch_backtracking(solver *S, ls_container *info, const orc_ls_config *cfg,
                                    double f_x, const double *df_x, double a_initial, double *o_phi,
                                    double *o_a, int64_t *o_evals, int *o_status)
{                                                                    /* geometric.jl:22-100 */
    const int64_t n = S->n;
    const double phi_0 = f_x;                                         /* :37 */
    if (!isfinite(phi_0)) {                                           /* :38-40 */
        *o_phi = phi_0; *o_a = 0.0; *o_evals = 0; *o_status = ORC_ACCEPTED_NON_FINITE_ITERATE;
        return;
    }
    const double dphi_0 = orc_dot(df_x, info->u, n);                  /* :42 */
    if (dphi_0 > 0.0) {                                               /* :43-45 */
        *o_phi = phi_0; *o_a = 0.0; *o_evals = 0; *o_status = ORC_NON_DESCENT_SEARCH_DIRECTION;
        return;
    }
    int64_t evals = 0;
    double a = a_initial;                                             /* :48 */
    if (!isfinite(a)) a = fabs(phi_0) / orc_dot(info->u, info->u, n); /* :49-52 */
    if (!isfinite(a)) a = 1.0;                                        /* :53-56 */
    double phi_a, dphi_a;
    const int flag = findfeasiblestepsize(S, info, &evals, &a, 0.5, 0.0, cfg->feasibility_max_iters,
                                          &phi_a, &dphi_a);           /* :58-70 */
    if (flag != ORC_FEASIBLE) {                                       /* :71-74 */
        *o_phi = phi_0; *o_a = 0.0; *o_evals = 0; *o_status = ORC_CANNOT_FIND_INITIAL_FEASIBLE_STEP;
        return;
    }
    eval_phi_dphi(S, info, a, &phi_a, &dphi_a);                       /* :77 (redundant re-evaluation) */
    evals += 1;
    note_margin(S, phi_0 - phi_a, -cfg->c1 * a * dphi_0, fmax(fabs(phi_0), fabs(phi_a)));
    const int valid = armijo_ok(cfg->c1, phi_a, a, phi_0, dphi_0);    /* :80 */
    geometricsearch(S, info, cfg, a, valid, evals, phi_a, phi_0, dphi_0, o_phi, o_a, o_evals,
                    o_status);                                        /* :82-97 */
}

/* ------------------------------------------------------------------ */
/* types.jl:134-151  updateresult!  (+ resizetrace! → iters_ran)       */
/* ------------------------------------------------------------------ */
static void updateresult(orc_results *ret, const double *x, const double *df_x, double f_x,
                         int64_t i, int status, int64_t n)
{
    ret->objective = f_x;
    if (ret->minimizer) memcpy(ret->minimizer, x, sizeof(double) * (size_t)n);
    if (ret->gradient) memcpy(ret->gradient, df_x, sizeof(double) * (size_t)n);
    ret->iters_ran = i;
    ret->status = status;
}

/* ------------------------------------------------------------------ */
/* optim.jl:6-171  minimizeobjective                                   */
/* ------------------------------------------------------------------ */
int orc_minimizeobjective(orc_fdf_t fdf, void *user, const double *x_initial, int64_t n,
                          const orc_cg_config *cfg, const orc_ls_config *ls, orc_results *ret)
{
    int e;
    if ((e = orc_check_cg_config(cfg)) != 0) return e;
    if ((e = orc_check_ls_config(ls)) != 0) return e;
    if (n < 1) return 5;

    const int64_t max_iters = cfg->max_iters;          /* :15 */
    const orc_beta_config *bcfg = &cfg->beta;          /* :16 */
    const int is_qn = bcfg->kind == ORC_BETA_LBFGS;
    const int is_bf = bcfg->kind == ORC_BETA_BROYDEN_FAMILY;
    if (is_bf && n > 4096) return 7;                   /* the dense n×n matrix of the reference */
    const size_t nb = sizeof(double) * (size_t)n;
    double *BF = NULL;                                 /* setupBroydenFamily: N×N filled with NaN (qn_flavours.jl:59-63) */
    if (is_bf) {
        BF = (double *)malloc(sizeof(double) * (size_t)(n * n));
        for (int64_t i = 0; i < n * n; ++i) BF[i] = NAN;
    }

    solver S;
    memset(&S, 0, sizeof(S));
    S.fdf = fdf; S.user = user; S.n = n; S.ret = ret;
    S.y = (double *)malloc(nb); S.tmp1 = (double *)malloc(nb); S.tmp2 = (double *)malloc(nb);

    double *df_x = (double *)malloc(nb);               /* :20 */
    double *x = (double *)malloc(nb);                  /* :21 */
    par_copy(x, x_initial, n);

    ret->log_len = 0;
    ret->snap_done = 0;
    double f_x = fdf(user, df_x, x, n);                /* :25 */
    S.total_evals = 1;
    double norm_df_x = orc_norm(df_x, n);              /* :26 */
    double norm_df_xp = NAN;                           /* :27 */
    double beta = 0.0;                                 /* :29 initializeβ → zeros(T,1) */
    int64_t fdf_evals_ran = -1;                        /* :30 */
    const double f_x0 = f_x;                           /* :31 */

    ret->objective = f_x;                              /* :34-41 */
    ret->iters_ran = 0;
    ret->status = ORC_INCOMPLETE;

    ls_container info;                                 /* :45 */
    info.xp = (double *)malloc(nb); info.df_xp = (double *)malloc(nb);
    info.x = (double *)malloc(nb);  info.u = (double *)malloc(nb);
    /* :46 initializeLineSearchContainer!  (cg_flavours.jl:22-35; for L-BFGS the
     * QN variant qn_flavours.jl:25-44 with B = I gives the same u = −g) */
    if (is_bf) broyden_updatedir(BF, info.u, df_x, n, S.tmp1); /* qn_flavours.jl:25-44 */
    else {
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) info.u[i] = -df_x[i];
    }
    par_copy(info.x, x, n);
    par_copy(info.xp, x, n);
    par_copy(info.df_xp, df_x, n);
    if (is_qn) lbfgs_init(&S.qn, bcfg->lbfgs_m, n);
    double a_initial = NAN;                            /* :47 */

    int status = ORC_MAX_ITERS_REACHED;
    int64_t iters = max_iters;
    for (int64_t it = 1; it <= max_iters; ++it) {      /* :50 */
        if (isfinite(f_x) && isfinite(norm_df_x)) {    /* :53 */
            if (norm_df_x < cfg->eps) {                /* :54 */
                status = (f_x <= f_x0) ? ORC_SUCCESS : ORC_INCREASING_OBJECTIVE; /* :56-78 */
                iters = it - 1;
                break;
            }
        }
        double f_xp, a_star;
        int ls_status;
        if (ls->kind == ORC_LS_STRONG_WOLFE_BISECTION)  /* :83-90 */
            linesearch_strong_wolfe(&S, &info, ls, f_x, df_x, a_initial, &f_xp, &a_star,
                                    &fdf_evals_ran, &ls_status);
        else if (ls->kind == ORC_LS_WOLFE_BISECTION)
            linesearch_wolfe_bisection(&S, &info, ls, f_x, df_x, a_initial, &f_xp, &a_star,
                                       &fdf_evals_ran, &ls_status);
        else
            linesearch_backtracking(&S, &info, ls, f_x, df_x, a_initial, &f_xp, &a_star,
                                    &fdf_evals_ran, &ls_status);
        a_initial = a_star;                            /* :92 */
        if (ls_status != ORC_SUCCESS) {                /* :93-104 */
            status = ls_status;
            iters = it - 1;
            break;
        }
        norm_df_xp = orc_norm(info.df_xp, n);          /* :107 */
        if (!isfinite(f_xp) || !isfinite(norm_df_xp)) { /* :108-121 */
            status = ORC_NON_FINITE_OBJECTIVE_OR_GRADIENT_PROPOSED;
            iters = it - 1;
            break;
        }
        if (is_qn)                                     /* :130-135 getβ */
            lbfgs_push(&S.qn, info.df_xp, df_x, info.u, a_star, n);
        else if (is_bf)
            broyden_getbeta(bcfg->mu, BF, info.df_xp, df_x, n);
        else
            beta = getbeta_impl(bcfg, info.df_xp, df_x, info.u, n, S.y, S.tmp1, S.tmp2);
        par_copy(x, info.xp, n);                       /* :136 */
        f_x = f_xp;                                    /* :138 */
        par_copy(df_x, info.df_xp, n);                 /* :139 */
        par_copy(info.x, x, n);                        /* :140 */
        norm_df_x = norm_df_xp;                        /* :141 */
        if (is_qn)                                     /* :145 updatedir! */
            lbfgs_updatedir(&S.qn, info.u, df_x, n);
        else if (is_bf)
            broyden_updatedir(BF, info.u, df_x, n, S.tmp1);
        else
            orc_updatedir(info.u, df_x, beta, n);
        if (cfg->trace_enabled && ret->trace_objective) { /* :152-159 updatetrace! */
            ret->trace_objective[it - 1] = f_x;
            ret->trace_grad_norm[it - 1] = norm_df_x;
            ret->trace_step_size[it - 1] = a_star;
            ret->trace_objective_evals[it - 1] = fdf_evals_ran;
        }
        if (ret->trace_time) ret->trace_time[it - 1] = orc_now();
        if (ret->snap_x && ret->snap_done < ret->nsnap && ret->snap_iters[ret->snap_done] == it) {
            par_copy(ret->snap_x + (size_t)ret->snap_done * (size_t)n, x, n);
            ret->snap_done++;
        }
    }
    updateresult(ret, x, df_x, f_x, iters, status, n); /* :162-170 and the early returns */
    ret->total_fdf_evals = S.total_evals;

    if (is_qn) lbfgs_free(&S.qn);
    free(BF);
    free(info.xp); free(info.df_xp); free(info.x); free(info.u);
    free(df_x); free(x);
    free(S.y); free(S.tmp1); free(S.tmp2);
    return 0;
}

/* ------------------------------------------------------------------ */
/* solve_system.jl  (Yuan, Wang & Sheng 2019: HZ-type CG for g(x) = 0)  */
/* ------------------------------------------------------------------ */
int orc_check_lss_config(const orc_lss_config *l)
{
    if (!l) return 1;
    if (!(0.0 < l->rho && l->rho < 1.0)) return 21; /* solve_system.jl:21 */
    if (!(l->rho > 0.0)) return 22;                 /* :22 (sic: tests ρ again, never σ) */
    if (!(l->s > 0.0)) return 23;                   /* :23 */
    return 0;
}

int64_t orc_lss_default_max_iters(double rho) /* round(Int, log(ρ, 1e-6)), ties to even like Julia */
{
    return (int64_t)nearbyint(log(1e-6) / log(rho));
}

int orc_solvesystem(orc_fdf_t fdf, void *user, const double *x_initial, int64_t n,
                    const orc_cg_config *cfg, const orc_lss_config *ls, orc_results *ret)
{
    int e;
    if ((e = orc_check_cg_config(cfg)) != 0) return e;
    if ((e = orc_check_lss_config(ls)) != 0) return e;
    if (n < 1) return 5;
    if (cfg->beta.kind >= ORC_BETA_LBFGS) return 6; /* BT <: CGβConfig (:69) */

    const int64_t max_iters = cfg->max_iters;          /* :75 */
    const size_t nb = sizeof(double) * (size_t)n;
    solver S;
    memset(&S, 0, sizeof(S));
    S.fdf = fdf; S.user = user; S.n = n; S.ret = ret;
    S.y = (double *)malloc(nb); S.tmp1 = (double *)malloc(nb); S.tmp2 = (double *)malloc(nb);

    double *df_x = (double *)malloc(nb);               /* :80 */
    double *x = (double *)malloc(nb);                  /* :81 */
    double *x_next = (double *)malloc(nb);             /* :82 */
    memcpy(x, x_initial, nb);
    memcpy(x_next, x_initial, nb);

    ret->log_len = 0;
    ret->_pad = 0;
    double f_x = fdf(user, df_x, x, n);                /* :86 */
    S.total_evals = 1;
    double norm_df_x = orc_norm(df_x, n);              /* :87 */
    double beta = 0.0;                                 /* :88 */
    ret->objective = f_x; ret->iters_ran = 0; ret->status = ORC_INCOMPLETE; /* :93-100 */

    ls_container info;                                 /* :104-105 */
    info.xp = (double *)malloc(nb); info.df_xp = (double *)malloc(nb);
    info.x = (double *)malloc(nb);  info.u = (double *)malloc(nb);
    for (int64_t i = 0; i < n; ++i) info.u[i] = -df_x[i];
    memcpy(info.x, x, nb); memcpy(info.xp, x, nb); memcpy(info.df_xp, df_x, nb);

    int status = ORC_MAX_ITERS_REACHED;
    int64_t iters = max_iters;
    const double *res_x = NULL, *res_g = NULL; /* what updateresult! copies out */
    for (int64_t it = 1; it <= max_iters; ++it) {      /* :109 */
        if (norm_df_x < cfg->eps) {                    /* :112 (no isfinite test here) */
            status = ORC_SUCCESS; iters = it - 1;
            break;
        }
        /* linesearch! (:29-56) */
        const double norm_u_sq = orc_dot(info.u, info.u, n);  /* :41 */
        double f_xp = NAN, norm_df_xp = NAN, a_star = NAN;
        int64_t evals_i = -1;
        int found = 0;
        for (int64_t i = 0; i < ls->max_iters; ++i) {  /* :43 */
            const double a = ls->s * pow(ls->rho, (double)i); /* :44  a0*ρ^i */
            double dphi;
            eval_phi_dphi(&S, &info, a, &f_xp, &dphi); /* :46 */
            norm_df_xp = orc_norm(info.df_xp, n);      /* :49 */
            if (!(-dphi < ls->sigma * a * norm_df_xp * norm_u_sq)) { /* :50 */
                a_star = a; evals_i = i; found = 1;    /* :52 */
                break;
            }
        }
        if (!found) {                                  /* :55 → UndefVarError in the reference */
            status = ORC_LINESEARCH_FAILED; iters = it - 1;
            ret->_pad = 1;
            break;
        }
        if (norm_df_xp < cfg->eps) {                   /* :145-166: the TRIAL point is the answer */
            if (cfg->trace_enabled && ret->trace_objective) {
                ret->trace_objective[it - 1] = f_xp;
                ret->trace_grad_norm[it - 1] = orc_norm(info.df_xp, n);
                ret->trace_step_size[it - 1] = a_star;
                ret->trace_objective_evals[it - 1] = evals_i;
            }
            status = ORC_SUCCESS; iters = it; f_x = f_xp;
            res_x = info.xp; res_g = info.df_xp;
            break;
        }
        /* updateiteratesolvesys!(x_next, df_xp, norm_df_xp, a_star, u)  :169-175, :239-253 */
        {
            const double m = a_star * orc_dot(info.df_xp, info.u, n) / (norm_df_xp * norm_df_xp); /* :248 */
            ORC_PAR
            for (int64_t i = 0; i < n; ++i) x_next[i] = x_next[i] + m * info.df_xp[i];          /* :250-252 */
        }
        const double f_x_next = fdf(user, info.df_xp, x_next, n); /* :177 */
        S.total_evals++;
        if (!isfinite(f_x_next) || !isfinite(orc_norm(info.df_xp, n))) { /* :178-191 */
            status = ORC_NON_FINITE_OBJECTIVE_OR_GRADIENT_PROPOSED; iters = it - 1;
            break;
        }
        { double *t = x; x = x_next; x_next = t; }     /* :194 */
        f_x = f_x_next;                                /* :195 */
        beta = getbeta_impl(&cfg->beta, info.df_xp, df_x, info.u, n, S.y, S.tmp1, S.tmp2); /* :199-204 */
        par_copy(df_x, info.df_xp, n);                 /* :205 */
        par_copy(info.x, x, n);                        /* :206 */
        norm_df_x = orc_norm(df_x, n);                 /* :207 */
        orc_updatedir(info.u, df_x, beta, n);          /* :210 */
        if (cfg->trace_enabled && ret->trace_objective) { /* :213-220 */
            ret->trace_objective[it - 1] = f_x;
            ret->trace_grad_norm[it - 1] = norm_df_x;
            ret->trace_step_size[it - 1] = a_star;
            ret->trace_objective_evals[it - 1] = evals_i;
        }
    }
    updateresult(ret, res_x ? res_x : x, res_g ? res_g : df_x, f_x, iters, status, n);
    ret->total_fdf_evals = S.total_evals;

    free(info.xp); free(info.df_xp); free(info.x); free(info.u);
    free(df_x); free(x); free(x_next);
    free(S.y); free(S.tmp1); free(S.tmp2);
    return 0;
}

/* optim.jl:173-208  minimizeobjectivererun */
int orc_minimizeobjectivererun(orc_fdf_t fdf, void *user, const double *x_initial, int64_t n,
                               const orc_cg_config *cfg, const orc_ls_config *ls,
                               const orc_cg_config *rerun_cfgs, const orc_ls_config *rerun_ls,
                               int npairs, orc_results *rets, int *nrets)
{
    int e = orc_minimizeobjective(fdf, user, x_initial, n, cfg, ls, &rets[0]); /* :183-188 */
    if (e) return e;
    int cnt = 1;
    for (int k = 0; k < npairs; ++k) {                  /* :191 */
        if (rets[cnt - 1].status != ORC_SUCCESS) {      /* :192 */
            e = orc_minimizeobjective(fdf, user, rets[cnt - 1].minimizer, n, &rerun_cfgs[k],
                                      &rerun_ls[k], &rets[cnt]); /* :195-200 */
            if (e) return e;
            cnt++;
        } else
            break;                                      /* :203 */
    }
    *nrets = cnt;
    return 0;
}

/* ------------------------------------------------------------------ */
/* objectives                                                          */
/* ------------------------------------------------------------------ */

/* examples/helpers/test_funcs.jl:3-12  boothfdf! */
double orc_fdf_booth(void *user, double *g, const double *p, int64_t n)
{
    (void)user; (void)n;
    const double x = p[0], y = p[1];
    const double t1 = x + 2 * y - 7, t2 = 2 * x + y - 5;
    const double f = t1 * t1 + t2 * t2;     /* :6 */
    g[0] = 2 * t1 + 2 * t2 * 2;             /* :8 */
    g[1] = 2 * t1 * 2 + 2 * t2;             /* :9 */
    return f;
}

/* separable quadratic f = ½ Σ D_i x_i²  (BASELINE configs 2 and 5; not in the reference) */
double orc_fdf_quad_diag(void *user, double *g, const double *x, int64_t n)
{
    const double *D = ((const orc_quad_params *)user)->D;
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) g[i] = D[i] * x[i];
#ifdef ORC_EXACT_SUMS
    {   /* the same terms 0.5·(g_i·x_i), each rounded as below, summed in twice the working precision */
        double *w = (double *)malloc(sizeof(double) * (size_t)n);
        ORC_PAR
        for (int64_t i = 0; i < n; ++i) w[i] = 0.5 * (g[i] * x[i]);
        const double f = orc_sum(w, n);
        free(w);
        return f;
    }
#endif
#ifdef _OPENMP
    if (n >= ORC_OMP_MIN) {
        double t = 0.0;
#pragma omp parallel for reduction(+ : t) schedule(static)
        for (int64_t i = 0; i < n; ++i) t += 0.5 * (g[i] * x[i]);
        return t;
    }
#endif
    /* f = Σ 0.5·(g_i·x_i), 8 partial sums like orc_dot */
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int k = 0; k < 8; ++k) s[k] += 0.5 * (g[i + k] * x[i + k]);
    double f = ((s[0] + s[4]) + (s[2] + s[6])) + ((s[1] + s[5]) + (s[3] + s[7]));
    for (; i < n; ++i) f += 0.5 * (g[i] * x[i]);
    return f;
}

/* extended (paired) Rosenbrock f = Σ_j 100(x_{2j} − x_{2j−1}²)² + (1 − x_{2j−1})²
 * (BASELINE config 3; value form of test_funcs.jl:50-57 restricted to disjoint pairs) */
double orc_fdf_rosenbrock_paired(void *user, double *g, const double *x, int64_t n)
{
    (void)user;
    const int64_t np = n / 2;
#if defined(ORC_EXACT_SUMS) || defined(_OPENMP)
#ifdef ORC_EXACT_SUMS
    const int par = 1;
    double *w = (double *)malloc(sizeof(double) * (size_t)(np > 0 ? np : 1));
#else
    const int par = n >= ORC_OMP_MIN;
    double *w = NULL;
#endif
    if (par) {
        double t = 0.0;
        ORC_PAR_SUM(t)
        for (int64_t j = 0; j < np; ++j) {
            const double a = x[2 * j], b = x[2 * j + 1];
            const double t1 = b - a * a, t2 = 1.0 - a;
            const double fj = 100.0 * (t1 * t1) + t2 * t2;
            g[2 * j] = -400.0 * (a * t1) - 2.0 * t2;
            g[2 * j + 1] = 200.0 * t1;
            if (w) w[j] = fj; else t += fj;
        }
        if (n & 1) g[n - 1] = 0.0;
#ifdef ORC_EXACT_SUMS
        t = orc_sum(w, np);
        free(w);
#endif
        return t;
    }
#endif
    double f0 = 0, f1 = 0, f2 = 0, f3 = 0;
    int64_t j = 0;
    for (; j < np; ++j) {
        const double a = x[2 * j], b = x[2 * j + 1];
        const double t1 = b - a * a, t2 = 1.0 - a;
        const double fj = 100.0 * (t1 * t1) + t2 * t2;
        g[2 * j] = -400.0 * (a * t1) - 2.0 * t2;
        g[2 * j + 1] = 200.0 * t1;
        switch (j & 3) { case 0: f0 += fj; break; case 1: f1 += fj; break;
                         case 2: f2 += fj; break; default: f3 += fj; }
    }
    if (n & 1) g[n - 1] = 0.0; /* odd tail element does not enter f */
    return (f0 + f2) + (f1 + f3);
}

/* chained Rosenbrock, value exactly as test_funcs.jl:50-57; gradient derived here */
double orc_fdf_rosenbrock_chained(void *user, double *g, const double *x, int64_t n)
{
    (void)user;
    double f = 0.0;
#ifdef ORC_EXACT_SUMS
    double *w = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
#endif
    for (int64_t i = 0; i < n; ++i) g[i] = 0.0;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const double t2 = 1.0 - x[i], t1 = x[i + 1] - x[i] * x[i];
#ifdef ORC_EXACT_SUMS
        w[i] = t2 * t2 + 100.0 * (t1 * t1);
#else
        f += t2 * t2 + 100.0 * (t1 * t1); /* :54 */
#endif
        g[i] += -2.0 * t2 - 400.0 * (x[i] * t1);
        g[i + 1] += 200.0 * t1;
    }
#ifdef ORC_EXACT_SUMS
    f = orc_sum(w, n > 0 ? n - 1 : 0);
    free(w);
#endif
    return f;
}

/* f = log Σ exp(x_i) + ½λ‖x‖²  (BASELINE config 4; not in the reference) */
double orc_fdf_lse(void *user, double *g, const double *x, int64_t n)
{
    const double lambda = ((const orc_lse_params *)user)->lambda;
    double m = -INFINITY;
    ORC_PAR_MAX(m)
    for (int64_t i = 0; i < n; ++i) m = x[i] > m ? x[i] : m;
    double s = 0.0;
#ifdef ORC_EXACT_SUMS
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) g[i] = exp(x[i] - m);
    s = orc_sum(g, n);
#else
    ORC_PAR_SUM(s)
    for (int64_t i = 0; i < n; ++i) {
        g[i] = exp(x[i] - m);
        s += g[i];
    }
#endif
    const double xx = orc_dot(x, x, n);
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) g[i] = g[i] / s + lambda * x[i];
    return (m + log(s)) + 0.5 * lambda * xx;
}

/* ------------------------------------------------------------------ */
/* counter-based RNG: splitmix64 finaliser of (seed XOR index)         */
/* ------------------------------------------------------------------ */
double orc_uniform(uint64_t seed, uint64_t index)
{
    uint64_t z = (seed ^ index) + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

void orc_fill_uniform(double *v, int64_t offset, int64_t n, uint64_t seed, double lo, double hi)
{
    for (int64_t i = 0; i < n; ++i)
        v[i] = lo + (hi - lo) * orc_uniform(seed, (uint64_t)(offset + i));
}

void orc_fill_quad_diag(double *D, int64_t offset, int64_t n, uint64_t seed, double lo, double hi)
{
    orc_fill_uniform(D, offset, n, seed, lo, hi);
}
