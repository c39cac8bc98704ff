/* The drop-in boundary from plain C: minimizeobjective (optim.jl:6-171) on the Booth function through
 * include/cgo.h — what any FFI (Julia ccall, cgo, JNI …) binds.
 *
 *   gcc -std=c11 -Iinclude examples/booth_min.c -Lconjugategradientoptim.jl_amd/lib -lcgo_hip \
 *       -Wl,-rpath,$PWD/conjugategradientoptim.jl_amd/lib -o /tmp/booth_min && /tmp/booth_min
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "cgo.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != CGO_OK) { \
    fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, cgo_last_error()); return 1; } } while (0)

int main(void) {
    cgo_ctx *ctx = NULL;
    cgo_objective *fdf = NULL;
    CHECK(cgo_ctx_create(0, &ctx));                                   /* no GPU → CGO_ENODEV: there is no CPU path */
    CHECK(cgo_objective_create(ctx, CGO_OBJ_BOOTH, 2, 0, 2, &fdf));

    cgo_cg_config cfg;                                                /* setupCGConfig(1e-5, HagerZhang(), EnableTrace(); max_iters = 1000) */
    memset(&cfg, 0, sizeof cfg);
    cfg.eps = 1e-5; cfg.beta.kind = CGO_BETA_HAGER_ZHANG; cfg.max_iters = 1000; cfg.trace_enabled = 1;
    cgo_ls_config ls;                                                 /* setupStrongWolfeBisection(1e-5, 0.8; growth 2, 1000, 100) */
    memset(&ls, 0, sizeof ls);
    ls.kind = CGO_LS_STRONG_WOLFE_BISECTION; ls.c1 = 1e-5; ls.c2 = 0.8; ls.a_max_growth_factor = 2.0;
    ls.max_iters = 1000; ls.zoom_max_iters = 100;
    CHECK(cgo_check_cg_config(&cfg));
    CHECK(cgo_check_ls_config(&ls));

    const double x0[2] = {0.43, 1.23};
    double x[2], g[2], tf[1000], tg[1000], ts[1000];
    int64_t te[1000];
    cgo_results r;
    memset(&r, 0, sizeof r);
    r.minimizer = x; r.gradient = g;
    r.trace_objective = tf; r.trace_grad_norm = tg; r.trace_step_size = ts; r.trace_objective_evals = te;
    CHECK(cgo_minimize(ctx, fdf, x0, &cfg, &ls, &r));

    int64_t evals = 0;
    for (int64_t i = 0; i < r.iters_ran; ++i) evals += te[i];
    printf("status %s after %lld iterations (%lld objective evaluations, %lld launches)\n",
           cgo_status_name(r.status), (long long)r.iters_ran, (long long)evals, (long long)r.total_launches);
    printf("minimizer (%.9f, %.9f)  objective %.3e  |gradient| %.3e\n", x[0], x[1], r.objective, hypot(g[0], g[1]));
    const int ok = r.status == CGO_SUCCESS && fabs(x[0] - 1.0) < 1e-4 && fabs(x[1] - 3.0) < 1e-4;
    CHECK(cgo_objective_destroy(fdf));
    CHECK(cgo_ctx_destroy(ctx));
    return ok ? 0 : 2;
}
