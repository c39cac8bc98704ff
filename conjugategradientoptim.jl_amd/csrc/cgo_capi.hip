// cgo_capi.hip — extern "C" boundary (include/cgo.h).  No C++ types or
// exceptions cross it; every entry point catches and converts to an error code.
#include <sys/mman.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "cgo_hip_backend.hpp"
#include "cgo_build_id.inc"

using namespace cgo;

// Lifetimes: an objective keeps its context alive and a solver keeps its objective alive, whatever order the host
// destroys the handles in (a garbage-collected host — Python at interpreter exit, Julia finalizers — gives no order).
// cgo_*_destroy drops the HOST's reference; the object goes when the last reference does.
struct cgo_ctx { HipCtx c; int refs = 1; cgo_solver_policy defpol; bool has_defpol = false; };
struct cgo_objective { HipObjective o; cgo_ctx *owner = nullptr; int refs = 1; };
static void ctx_unref(cgo_ctx *c) { if (c && --c->refs == 0) delete c; }
static void obj_unref(cgo_objective *o) {
    if (o && --o->refs == 0) {
        cgo_ctx *c = o->owner;
        if (c) (void)hipSetDevice(c->c.device);
        delete o;
        ctx_unref(c);
    }
}
struct cgo_solver {
    cgo_ctx *ctx;
    cgo_objective *obj;
    HipBackend *be;
    Solver *sv;
    cgo_solver_policy pol;   // what it runs with (cgo_solver_get_policy)
    ~cgo_solver() { delete sv; delete be; if (obj && obj->o.users > 0) obj->o.users--; }
};

#define API_GUARD_BEGIN try {
#define API_GUARD_END                                                        \
    } catch (const std::bad_alloc &) { set_error("out of host memory"); return CGO_ENOMEM; } \
    catch (const std::exception &e) { set_error(std::string("internal: ") + e.what()); return CGO_EINVAL; } \
    catch (...) { set_error("internal: unknown exception"); return CGO_EINVAL; }

#define REQUIRE(cond, msg) do { if (!(cond)) { set_error(msg); return CGO_EINVAL; } } while (0)
#define HIPCHK2(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { \
    set_error(std::string("HIP error: ") + hipGetErrorString(e__) + " at " #expr); return CGO_EHIP; } } while (0)

extern "C" {

int cgo_version(void) { return CGO_VERSION; }
const char *cgo_build_id(void) { return CGO_BUILD_ID; }
const char *cgo_last_error(void) { return get_error(); }
const char *cgo_status_name(int32_t s) { return status_name(s); }
const char *cgo_kernel_kind_name(int32_t k) {
    if (k == 100) return "dir";
    if (k == 101) return "beta_partials";
    return kernel_kind_name(k);
}
int cgo_num_kernel_kinds(void) { return KK_COUNT; }

int cgo_device_count(int32_t *count) {
    REQUIRE(count, "null count");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *count = (e == hipSuccess) ? c : 0;
    return CGO_OK;
}

int cgo_check_cg_config(const cgo_cg_config *cfg) {
    std::string why;
    int rc = check_cg_config(cfg, why);
    if (rc) set_error(why);
    return rc;
}
int cgo_check_ls_config(const cgo_ls_config *ls) {
    std::string why;
    int rc = check_ls_config(ls, why);
    if (rc) set_error(why);
    return rc;
}

// ---- ctx ------------------------------------------------------------------
int cgo_ctx_create(int32_t device, cgo_ctx **out) {
    API_GUARD_BEGIN
    REQUIRE(out, "null out");
    *out = nullptr;
    cgo_ctx *c = new cgo_ctx();
    int rc = c->c.init(device);
    if (rc) { std::string keep = get_error(); delete c; set_error(keep); return rc; }
    *out = c;
    return CGO_OK;
    API_GUARD_END
}

int cgo_ctx_destroy(cgo_ctx *ctx) {
    API_GUARD_BEGIN
    ctx_unref(ctx);
    return CGO_OK;
    API_GUARD_END
}

int cgo_comm_unique_id(void *out128) {
    API_GUARD_BEGIN
    REQUIRE(out128, "null out");
    return rccl_unique_id(out128);
    API_GUARD_END
}

int cgo_ctx_set_comm_rccl(cgo_ctx *ctx, int32_t rank, int32_t world, const void *uid) {
    API_GUARD_BEGIN
    REQUIRE(ctx && uid, "null argument");
    REQUIRE(world >= 1 && world <= 64 && rank >= 0 && rank < world, "bad rank/world");
    Comm *c = make_rccl_comm(&ctx->c, rank, world, uid);
    if (!c) return CGO_ECOMM;
    ctx->c.comm.reset(c);
    return ctx->c.ensure_gather();
    API_GUARD_END
}

int cgo_ctx_set_comm_callback(cgo_ctx *ctx, int32_t rank, int32_t world, cgo_allgather_fn fn, void *user) {
    API_GUARD_BEGIN
    REQUIRE(ctx && fn, "null argument");
    REQUIRE(world >= 1 && world <= 64 && rank >= 0 && rank < world, "bad rank/world");
    ctx->c.comm.reset(make_callback_comm(rank, world, fn, user));
    return ctx->c.ensure_gather();
    API_GUARD_END
}

int cgo_ctx_set_comm_shm(cgo_ctx *ctx, int32_t rank, int32_t world, const char *name, int32_t create) {
    API_GUARD_BEGIN
    REQUIRE(ctx && name, "null argument");
    REQUIRE(world >= 1 && world <= 64 && rank >= 0 && rank < world, "bad rank/world");
    Comm *c = make_shm_comm(&ctx->c, rank, world, name, create);
    if (!c) return CGO_ECOMM;
    ctx->c.comm.reset(c);
    return CGO_OK;
    API_GUARD_END
}

int cgo_ctx_comm_connect_devices(cgo_ctx *ctx, int32_t *connected) {
    API_GUARD_BEGIN
    REQUIRE(ctx, "null argument");
    Comm *c = ctx->c.comm.get();
    HIPCHK2(hipSetDevice(ctx->c.device));
    const int ok = c ? c->connect_devices() : 0;
    if (connected) *connected = ok;
    return CGO_OK;
    API_GUARD_END
}

int cgo_ctx_comm_info(cgo_ctx *ctx, int32_t *kind, int32_t *rank, int32_t *world, int32_t *ranks_seen) {
    API_GUARD_BEGIN
    REQUIRE(ctx, "null argument");
    Comm *c = ctx->c.comm.get();
    if (kind) *kind = c ? c->kind() : 0;
    if (rank) *rank = ctx->c.rank();
    if (world) *world = ctx->c.world();
    if (ranks_seen) *ranks_seen = c ? c->ranks_seen() : 1;
    return CGO_OK;
    API_GUARD_END
}

int cgo_ctx_exchange_stats(cgo_ctx *ctx, int64_t *exchanges, double *peer_wait_us, double *device_exchange_us, int32_t reset) {
    API_GUARD_BEGIN
    REQUIRE(ctx, "null argument");
    HipCtx &c = ctx->c;
    if (c.xev_pending) { (void)hipSetDevice(c.device); (void)hipEventSynchronize(c.xev1); c.xch_collect(); }
    if (exchanges) *exchanges = c.xch_count;
    if (peer_wait_us) *peer_wait_us = c.xch_peer_wait_ns * 1e-3;
    if (device_exchange_us) *device_exchange_us = c.xch_dev_n ? c.xch_dev_ms * 1e3 / (double)c.xch_dev_n : 0.0;
    if (reset) { c.xch_count = 0; c.xch_peer_wait_ns = 0; c.xch_dev_ms = 0; c.xch_dev_n = 0; }
    return CGO_OK;
    API_GUARD_END
}

int cgo_rccl_available(void) { return rccl_available() ? 1 : 0; }

int cgo_shm_unlink(const char *name) {
    API_GUARD_BEGIN
    REQUIRE(name, "null argument");
    shm_unlink(name);
    return CGO_OK;
    API_GUARD_END
}

// ---- objective --------------------------------------------------------------
int cgo_objective_create(cgo_ctx *ctx, int32_t kind, int64_t n_global, int64_t offset,
                         int64_t n_local, cgo_objective **out) {
    API_GUARD_BEGIN
    REQUIRE(ctx && out, "null argument");
    *out = nullptr;
    REQUIRE((kind >= CGO_OBJ_QUAD_DIAG && kind <= CGO_OBJ_LSE) || kind == CGO_OBJ_ROSENBROCK_CHAINED,
            "unknown objective kind (user objectives: cgo_objective_create_from_source, closures: cgo_objective_create_callback)");
    REQUIRE(n_local >= 1 && offset >= 0 && offset + n_local <= n_global, "bad shard extents");
    if (kind == CGO_OBJ_ROSENBROCK_PAIRED)
        REQUIRE((offset % 2 == 0) && (n_local % 2 == 0), "paired Rosenbrock: shard offset and length must be even");
    if (kind == CGO_OBJ_BOOTH) REQUIRE(n_global == 2 && n_local == 2 && offset == 0, "Booth is 2-dimensional");
    if (kind == CGO_OBJ_ROSENBROCK_CHAINED)   // (the odd tail element of an odd N lives on the rank that ends the global vector)
        REQUIRE((offset % 2 == 0) && n_global >= 2 && n_local >= 2 && ((n_local % 2 == 0) || offset + n_local == n_global),
                "chained Rosenbrock: shard offsets must be even, and only the last shard may have odd length");
    if (kind == CGO_OBJ_QUAD_DIAG && ctx->c.world() > 1)
        REQUIRE(offset % 2 == 0, "shard offset must be even");
    cgo_objective *o = new cgo_objective();
    o->owner = ctx; ctx->refs++;
    o->o.ctx = &ctx->c; o->o.kind = kind; o->o.n_global = n_global; o->o.offset = offset; o->o.n_local = n_local;
    if (o->o.uses_param()) {
        HIPCHK2(hipSetDevice(ctx->c.device));
        int rc = o->o.p0.alloc((size_t)n_local);
        if (rc) { obj_unref(o); return rc; }
    }
    *out = o;
    return CGO_OK;
    API_GUARD_END
}

int cgo_objective_create_from_source(cgo_ctx *ctx, const char *source, int32_t has_param, int64_t n_global,
                                     int64_t offset, int64_t n_local, cgo_objective **out) {
    API_GUARD_BEGIN
    REQUIRE(ctx && source && out, "null argument");
    *out = nullptr;
    REQUIRE(n_local >= 1 && offset >= 0 && offset + n_local <= n_global, "bad shard extents");
    REQUIRE(offset % 2 == 0, "shard offset must be even");
    std::shared_ptr<RtcModule> mod;
    std::string log;
    int rc = rtc_compile_objective(ctx->c.device, source, has_param != 0, mod, log);
    if (rc) { set_error("user objective did not compile:\n" + log); return rc; }
    cgo_objective *o = new cgo_objective();
    o->owner = ctx; ctx->refs++;
    o->o.ctx = &ctx->c; o->o.kind = CGO_OBJ_USER; o->o.n_global = n_global; o->o.offset = offset; o->o.n_local = n_local;
    o->o.rtc = mod; o->o.user_has_param = has_param != 0;
    if (o->o.uses_param()) {
        HIPCHK2(hipSetDevice(ctx->c.device));
        rc = o->o.p0.alloc((size_t)n_local);
        if (rc) { obj_unref(o); return rc; }
    }
    *out = o;
    return CGO_OK;
    API_GUARD_END
}

int cgo_objective_create_callback(cgo_ctx *ctx, cgo_fdf_fn fn, void *user, int64_t n_global, int64_t offset,
                                  int64_t n_local, cgo_objective **out) {
    API_GUARD_BEGIN
    REQUIRE(ctx && fn && out, "null argument");
    *out = nullptr;
    REQUIRE(n_local >= 1 && offset >= 0 && offset + n_local <= n_global, "bad shard extents");
    HIPCHK2(hipSetDevice(ctx->c.device));
    cgo_objective *o = new cgo_objective();
    o->owner = ctx; ctx->refs++;
    o->o.ctx = &ctx->c; o->o.kind = CGO_OBJ_HOST; o->o.n_global = n_global; o->o.offset = offset; o->o.n_local = n_local;
    o->o.host_fn = fn; o->o.host_user = user;
    const size_t bytes = sizeof(double) * (size_t)n_local;
    if (hipHostMalloc((void **)&o->o.host_x, bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&o->o.host_g, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        obj_unref(o);
        set_error("could not pin host staging buffers for the callback objective");
        return CGO_ENOMEM;
    }
    *out = o;
    return CGO_OK;
    API_GUARD_END
}

int cgo_objective_destroy(cgo_objective *obj) {
    API_GUARD_BEGIN
    obj_unref(obj);
    return CGO_OK;
    API_GUARD_END
}

int cgo_objective_set_param_host(cgo_objective *obj, int32_t slot, const double *host) {
    API_GUARD_BEGIN
    REQUIRE(obj && host, "null argument");
    REQUIRE(slot == 0 && obj->o.uses_param(), "objective has no such parameter vector");
    HipCtx *c = obj->o.ctx;
    HIPCHK2(hipSetDevice(c->device));
    HIPCHK2(hipMemcpyAsync(obj->o.p0.p, host, sizeof(double) * (size_t)obj->o.n_local, hipMemcpyHostToDevice, c->stream));
    HIPCHK2(hipStreamSynchronize(c->stream));
    obj->o.p0_set = true;
    return CGO_OK;
    API_GUARD_END
}

int cgo_objective_fill_param(cgo_objective *obj, int32_t slot, int32_t fill_kind, uint64_t seed,
                             double lo, double hi) {
    API_GUARD_BEGIN
    REQUIRE(obj, "null argument");
    REQUIRE(slot == 0 && obj->o.uses_param(), "objective has no such parameter vector");
    REQUIRE(fill_kind >= 0 && fill_kind <= 2, "unknown fill kind");
    HipCtx *c = obj->o.ctx;
    HIPCHK2(hipSetDevice(c->device));
    if (int rc = fill_device(c, obj->o.p0.p, obj->o.n_local, obj->o.offset, fill_kind, seed, lo, hi)) return rc;
    HIPCHK2(hipStreamSynchronize(c->stream));
    obj->o.p0_set = true;
    return CGO_OK;
    API_GUARD_END
}

int cgo_objective_set_scalar(cgo_objective *obj, int32_t slot, double value) {
    API_GUARD_BEGIN
    REQUIRE(obj && slot == 0, "bad argument");
    obj->o.s0 = value;
    return CGO_OK;
    API_GUARD_END
}

int cgo_objective_set_cost_class(cgo_objective *obj, int32_t cost_class) {
    API_GUARD_BEGIN
    REQUIRE(obj && (cost_class == 0 || cost_class == 1), "bad argument");
    REQUIRE(obj->o.kind == CGO_OBJ_USER, "built-in objectives carry their own cost class");
    obj->o.user_cheap = cost_class == 1;
    return CGO_OK;
    API_GUARD_END
}

int cgo_objective_eval_host(cgo_objective *obj, const double *x, double *g, double *f) {
    API_GUARD_BEGIN
    REQUIRE(obj && x && f, "null argument");
    if (obj->o.host_closure()) {   // the closure IS the host form
        std::vector<double> tmp;
        if (!g) { tmp.resize((size_t)obj->o.n_local); g = tmp.data(); }
        *f = obj->o.host_fn(obj->o.host_user, g, x, obj->o.n_local);
        return CGO_OK;
    }
    return HipBackend::run_eval(&obj->o, x, g, f);
    API_GUARD_END
}

// ---- solver policy ------------------------------------------------------------
void cgo_solver_policy_init(cgo_solver_policy *p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->size = (int32_t)sizeof(*p);
    p->resident = p->controller_depth = p->controller_graph = p->controller_fused = -1;
    p->fused_tail = p->strict_tail = p->placement_search = -1;
    p->lbfgs_fuse_grad = p->lbfgs_fuse_trial = p->lse_fixed_reference = -1;
}

int cgo_ctx_set_default_policy(cgo_ctx *ctx, const cgo_solver_policy *policy) {
    API_GUARD_BEGIN
    REQUIRE(ctx, "null argument");
    if (!policy) { ctx->has_defpol = false; return CGO_OK; }
    REQUIRE(policy->size == (int32_t)sizeof(cgo_solver_policy), "cgo_solver_policy.size does not match this library (call cgo_solver_policy_init first)");
    ctx->defpol = *policy; ctx->has_defpol = true;
    return CGO_OK;
    API_GUARD_END
}

// explicit argument > context default > CGO_* experiment override > library policy, field by field
static int env_tri(const char *name) { const char *e = getenv(name); return e ? (e[0] != '0' ? 1 : 0) : -1; }
static int resolve_policy(cgo_ctx *ctx, const cgo_solver_policy *arg, cgo_solver_policy &r, std::string &why) {
    cgo_solver_policy lib; cgo_solver_policy_init(&lib);
    const cgo_solver_policy *defp = ctx->has_defpol ? &ctx->defpol : nullptr;
    for (const cgo_solver_policy *p : {arg, defp})
        if (p && p->size != (int32_t)sizeof(cgo_solver_policy)) { why = "cgo_solver_policy.size does not match this library (call cgo_solver_policy_init first)"; return CGO_EINVAL; }
    r = lib;
    const cgo_solver_policy *layers[2] = {defp, arg};   // later wins
    for (const cgo_solver_policy *p : layers) {
        if (!p) continue;
        if (p->points) r.points = p->points;
        if (p->resident >= 0) r.resident = p->resident;
        if (p->controller_depth >= 0) r.controller_depth = p->controller_depth;
        if (p->controller_graph >= 0) r.controller_graph = p->controller_graph;
        if (p->controller_fused >= 0) r.controller_fused = p->controller_fused;
        if (p->stored_gradient) r.stored_gradient = p->stored_gradient;
        if (p->fused_tail >= 0) r.fused_tail = p->fused_tail;
        if (p->strict_tail >= 0) r.strict_tail = p->strict_tail;
        if (p->placement_search >= 0) r.placement_search = p->placement_search;
        if (p->placement_stages) r.placement_stages = p->placement_stages;
        if (p->placement_max_bytes) r.placement_max_bytes = p->placement_max_bytes;
        if (p->lbfgs_form) r.lbfgs_form = p->lbfgs_form;
        if (p->lbfgs_fuse_grad >= 0) r.lbfgs_fuse_grad = p->lbfgs_fuse_grad;
        if (p->lbfgs_fuse_trial >= 0) r.lbfgs_fuse_trial = p->lbfgs_fuse_trial;
        if (p->lse_fixed_reference >= 0) r.lse_fixed_reference = p->lse_fixed_reference;
        if (p->resident_points) r.resident_points = p->resident_points;
        if (p->resident_chunk) r.resident_chunk = p->resident_chunk;
        if (p->hbm_stream_bytes > 0.0) r.hbm_stream_bytes = p->hbm_stream_bytes;
    }
    if (!(r.points == 0 || r.points == 1 || r.points == 3 || r.points == 5 || r.points == 7)) { why = "cgo_solver_policy.points must be 0, 1, 3, 5 or 7"; return CGO_EINVAL; }
    if (r.lbfgs_form < 0 || r.lbfgs_form > 4) { why = "cgo_solver_policy.lbfgs_form must be 0 … 4"; return CGO_EINVAL; }
    if (!(r.resident_points == 0 || r.resident_points == 1 || r.resident_points == 3 || r.resident_points == 7)) { why = "cgo_solver_policy.resident_points must be 0, 1, 3 or 7"; return CGO_EINVAL; }
    if (r.placement_stages < 0 || r.placement_stages > 3 || r.placement_max_bytes < 0) { why = "cgo_solver_policy.placement_* out of range"; return CGO_EINVAL; }
    // experiment overrides, only where nobody chose (the CGO_MULTI*_MIN_N thresholds stay with make_solver: they are not a point count)
    if (r.resident < 0) r.resident = env_tri("CGO_RESIDENT");
    if (r.controller_depth < 0) { if (const char *e = getenv("CGO_CTL_DEPTH")) r.controller_depth = std::max(0, atoi(e)); }
    if (r.controller_graph < 0) r.controller_graph = env_tri("CGO_CTL_GRAPH");
    if (r.controller_fused < 0) r.controller_fused = env_tri("CGO_CTL_FUSED");
    if (!r.stored_gradient) { const char *e = getenv("CGO_STORED_G"); r.stored_gradient = (e && e[0] == '1') ? 1 : 0; }
    if (r.placement_search < 0) r.placement_search = env_tri("CGO_PLACE_TUNE");
    if (!r.placement_stages) { if (const char *e = getenv("CGO_PLACE_STAGES")) { const int v = atoi(e); if (v >= 1 && v <= 3) r.placement_stages = v; } }
    if (!r.lbfgs_form) {
        const char *tl = getenv("CGO_LBFGS_TWO_LOOP"), *sp = getenv("CGO_LBFGS_SPEC");
        if (tl && tl[0] == '1') r.lbfgs_form = 4;
        else if (sp && sp[0] == '0') r.lbfgs_form = 3;
        else if (sp && sp[0] == '1') r.lbfgs_form = 2;
    }
    if (r.lbfgs_fuse_grad < 0) r.lbfgs_fuse_grad = env_tri("CGO_LBFGS_FUSE_GRAD");
    if (r.lbfgs_fuse_trial < 0) r.lbfgs_fuse_trial = env_tri("CGO_LBFGS_FUSE_TRIAL");
    if (r.lse_fixed_reference < 0) r.lse_fixed_reference = env_tri("CGO_LSE_REF");
    if (!r.resident_points) { if (const char *e = getenv("CGO_RES_POINTS")) { const int v = atoi(e); if (v == 1 || v == 3 || v == 7) r.resident_points = v; } }
    if (!r.resident_chunk) { if (const char *e = getenv("CGO_RES_CHUNK")) { const long long v = atoll(e); if (v >= 2 && v < (1LL << 30)) r.resident_chunk = (int32_t)(v & ~1LL); } }
    if (!(r.hbm_stream_bytes > 0.0)) { if (const char *e = getenv("CGO_BIG_BYTES")) { const double v = atof(e); if (v > 0.0) r.hbm_stream_bytes = v; } }
    // library values of the switches nobody touched (the tri-states the backend reads as plain booleans)
    if (r.controller_graph < 0) r.controller_graph = 0;
    if (r.controller_fused < 0) r.controller_fused = 1;
    if (r.lbfgs_fuse_grad < 0) r.lbfgs_fuse_grad = 1;
    if (r.lbfgs_fuse_trial < 0) r.lbfgs_fuse_trial = 1;
    if (r.lse_fixed_reference < 0) r.lse_fixed_reference = 1;
    if (r.placement_search < 0) r.placement_search = 0;   // OPT-IN (round 4): a measured heuristic that costs 10–450 ms and transient memory per solver
                                                          // and finds a faster buffer triple in 5 of 8 processes — the caller decides (bench.py does)
    return CGO_OK;
}

int cgo_solver_get_policy(cgo_solver *s, cgo_solver_policy *out) {
    API_GUARD_BEGIN
    REQUIRE(s && out, "null argument");
    *out = s->pol;
    out->points = s->be->policy_points();
    out->resident = s->be->resident_enabled() ? 1 : 0;
    out->controller_depth = s->be->ctl_depth_setting();
    out->fused_tail = s->ctx->c.fused_tail ? 1 : 0;
    out->strict_tail = s->ctx->c.tail_strict ? 1 : 0;
    return CGO_OK;
    API_GUARD_END
}

// ---- solver -----------------------------------------------------------------
static int make_solver(cgo_ctx *ctx, cgo_objective *obj, const cgo_cg_config *cfg, const cgo_ls_config *ls,
                       const cgo_lss_config *lss, const cgo_solver_policy *policy, cgo_solver **out) {
    *out = nullptr;
    REQUIRE(obj->o.ctx == &ctx->c, "objective belongs to another ctx");
    std::string why;
    if (int rc = check_cg_config(cfg, why)) { set_error(why); return rc; }
    if (ls) { if (int rc = check_ls_config(ls, why)) { set_error(why); return rc; } }
    else {
        if (int rc = check_lss_config(lss, why)) { set_error(why); return rc; }
        REQUIRE(cfg->beta.kind < CGO_BETA_LBFGS, "solvesystem takes a CGβConfig (solve_system.jl:69), not a QNβConfig");
        REQUIRE(!obj->o.two_phase() && !obj->o.host_closure() && obj->o.kind != CGO_OBJ_ROSENBROCK_CHAINED,
                "solvesystem needs an element-wise device objective (k_cg kernel family)");
    }
    const bool chain = obj->o.kind == CGO_OBJ_ROSENBROCK_CHAINED;
    if (chain) REQUIRE(cfg->beta.kind != CGO_BETA_LBFGS, "the chained (stencil) Rosenbrock objective runs on the gradient-free CG kernels: CG β kinds only");
    cgo_solver_policy pol;
    { std::string pw; if (int rc = resolve_policy(ctx, policy, pol, pw)) { set_error(pw); return rc; } }
    cgo_solver *s = new cgo_solver();
    s->ctx = ctx; s->obj = obj; s->pol = pol;
    obj->refs++;
    obj->o.users++;
    s->be = new HipBackend(&ctx->c, &obj->o);
    s->sv = nullptr;
    s->be->set_policy(pol);
    s->be->set_need_beta(cfg->beta.kind != CGO_BETA_LBFGS);
    // element-wise objective + CG β: gradient-free multi-point kernels (cgo_kernels_cg.hip.hpp);
    // policy.stored_gradient = 1 keeps the stored-gradient single-point family (A/B measurements)
    s->be->set_rmode(chain || (!obj->o.two_phase() && !obj->o.host_closure() && cfg->beta.kind != CGO_BETA_LBFGS && (!ls || !pol.stored_gradient)));
    int rc = s->be->alloc();   // after the family is known: the gradient-free family resides in x, u (+ D) only
    if (rc) { delete s; obj_unref(obj); return rc; }
    // How many trial steps a launch evaluates.  A saved launch is worth ≈ 15–25 µs at small n and a whole
    // pass over x,u,D at large n, so speculation pays at EVERY size (quadratic objective, 1 / 3 / 5 / 7 points,
    // it/s on MI355X: n = 1e4: 26.1k / 31.6k / 34.1k / 34.5k; 1e5: 24.3k / 29.2k / 33.7k / 33.4k;
    // 1e6: 22.6k / 27.4k / 23.4k / 24.1k; 3e6: 13.6k / 16.9k / 18.2k / 18.6k; 1e7: – / 9.2k / 9.7k / 9.5k;
    // 3e7: – / 3035 / 3140 / 3410; 1e8: – / 1037 / 1131 / 1213) as long as the extra FP64 work hides behind
    // the memory stream; the extended Rosenbrock kernel is VALU-bound beyond three points (1e7: 13.1k / 10.8k /
    // 8.6k; HZ + weak Wolfe there accepts most first trials, and 3 points only pay from n ≈ 3e6: 1 / 3 points at
    // 1e5: 41.6k / 36.0k, 1e6: 33.9k / 33.0k, 3e6: 22.0k / 23.1k).  Policy: the cheap built-in objectives use seven
    // points, except around n = 1e6 where the state just fits L2 + Infinity Cache and the wider rows cost more
    // than they save (three there); every other objective one point below n = 3e6 and three above.
    // Round 2, after the transpose-reduce tail and the fused reduction made a 24-slot row cheap: the cheap
    // class takes seven points at every size, every other objective THREE at every size — extended Rosenbrock, HZ + weak
    // Wolfe, host-driven, events off, 1 / 3 / 7 points: n = 1e3 56.0k / 73.6k / 63.6k it/s, 1e4 57.1k / 72.3k / 61.4k,
    // 1e5 52.6k / 68.9k / 58.1k, 1e6 44.8k / 54.7k / 44.2k, 3e6 32.2k / 36.5k / 29.5k (scripts/r02_points_rosen.sh).
    // CGO_MULTI_MIN_N / CGO_MULTI5_MIN_N / CGO_MULTI7_MIN_N override (and switch the 1e6 band off).
    const bool cheap = obj->o.kind == CGO_OBJ_QUAD_DIAG || obj->o.kind == CGO_OBJ_BOOTH ||
                       (obj->o.kind == CGO_OBJ_USER && obj->o.user_cheap);
    s->be->set_multi_min_n(0);
    s->be->set_multi5_min_n(cheap ? 0 : INT64_MAX);   // (the stencil objective is not in the cheap class: k_chain carries one or three points)
    s->be->set_multi7_min_n(cheap ? 0 : INT64_MAX);
    s->be->set_three_point_band(0, 0);   // round 1 kept three points for 5e5 ≤ n < 2e6; with the transpose-reduce tail seven win there too (n = 1e6: 45.1k vs 41.7k it/s)
    const char *mm = getenv("CGO_MULTI_MIN_N"), *m5 = getenv("CGO_MULTI5_MIN_N"), *m7 = getenv("CGO_MULTI7_MIN_N");
    if (chain) m5 = m7 = nullptr;   // the stencil launches evaluate one or three trial points, whatever the 5/7 knobs say
    if (mm || m5 || m7) s->be->set_three_point_band(0, 0);
    if (mm) s->be->set_multi_min_n(atoll(mm));
    if (m5) s->be->set_multi5_min_n(atoll(m5));
    if (m7) s->be->set_multi7_min_n(atoll(m7));
    if (pol.points) {   // an explicit point count: at every size (the stencil launches carry one or three)
        const int pts = chain ? std::min(pol.points, 3) : pol.points;
        s->be->set_three_point_band(0, 0);
        s->be->set_multi_min_n(pts >= 3 ? 0 : INT64_MAX);
        s->be->set_multi5_min_n(pts >= 5 ? 0 : INT64_MAX);
        s->be->set_multi7_min_n(pts >= 7 ? 0 : INT64_MAX);
    }
    // On-device line-search controller (cgo_ctl.hpp; since round 2 a whole armed round is ONE launch — tail_ctl).  It keeps
    // first-trial streaks on the device, which pays only while a launch is shorter than the host's turnaround: with
    // host-driven launches finishing their own sums too (finish_tail), extended Rosenbrock HZ + Wolfe without profiling
    // events runs, host-driven vs armed (median of three windows): n = 1e4 70–75k vs 85–87k it/s, 1e5 66–69k vs 74–75k,
    // 1e6 49–53k vs 51k (first window 47.5k vs 43.8k), 3e6 38.0k vs 35.4k, 1e7 17.4k vs 17.2k (gpurun_out/r02_cp).
    // Seven-point launches (the cheap class) gain nothing from it at any size (quadratic n = 1e6: 47.1k vs 45.7k).
    // Round 4 (after the armed launches stopped keeping their argument copy in scratch memory, and against today's host path;
    // profiles/r04_controller_scratch_fix.txt, events off, medians): n = 1e4 88–91k vs 90–97k, 1e5 76–77k vs 77–79k,
    // 3e5 64–65k vs 64k, 1e6 57–59k vs 55k, 3e6 38k vs 36k — armed up to n_local = 1e5 now (was 3e5).
    s->be->set_ctl_depth((ls && !cheap && !chain && s->be->policy_points() <= 3 && (ctx->c.world() == 1 || ctx->c.dev_exchange()) && obj->o.n_local <= 100000) ? 4 : 0);
    if (pol.controller_depth >= 0) s->be->set_ctl_depth(chain ? 0 : pol.controller_depth);  // 0: host drives every launch
    s->be->set_ctl_graph(pol.controller_graph != 0);                                       // 0: armed rounds kernel by kernel
    if (int prc = s->be->place()) { delete s; obj_unref(obj); return prc; }
    if (pol.resident >= 0) s->be->set_resident(pol.resident != 0);
    if (int prc = s->be->prepare_controller()) { delete s; obj_unref(obj); return prc; }   // the controller's blocks: now, not inside the first armed iteration
    s->sv = ls ? new Solver(s->be, *cfg, *ls) : new Solver(s->be, *cfg, *lss);
    *out = s;
    return CGO_OK;
}

int cgo_solver_create(cgo_ctx *ctx, cgo_objective *obj, const cgo_cg_config *cfg,
                      const cgo_ls_config *ls, cgo_solver **out) {
    API_GUARD_BEGIN
    REQUIRE(ctx && obj && cfg && ls && out, "null argument");
    return make_solver(ctx, obj, cfg, ls, nullptr, nullptr, out);
    API_GUARD_END
}

int cgo_solver_create_ex(cgo_ctx *ctx, cgo_objective *obj, const cgo_cg_config *cfg, const cgo_ls_config *ls,
                         const cgo_solver_policy *policy, cgo_solver **out) {
    API_GUARD_BEGIN
    REQUIRE(ctx && obj && cfg && ls && out, "null argument");
    return make_solver(ctx, obj, cfg, ls, nullptr, policy, out);
    API_GUARD_END
}

int cgo_solver_create_sys(cgo_ctx *ctx, cgo_objective *obj, const cgo_cg_config *cfg,
                          const cgo_lss_config *ls, cgo_solver **out) {
    API_GUARD_BEGIN
    REQUIRE(ctx && obj && cfg && ls && out, "null argument");
    return make_solver(ctx, obj, cfg, nullptr, ls, nullptr, out);
    API_GUARD_END
}

int cgo_solver_create_sys_ex(cgo_ctx *ctx, cgo_objective *obj, const cgo_cg_config *cfg, const cgo_lss_config *ls,
                             const cgo_solver_policy *policy, cgo_solver **out) {
    API_GUARD_BEGIN
    REQUIRE(ctx && obj && cfg && ls && out, "null argument");
    return make_solver(ctx, obj, cfg, nullptr, ls, policy, out);
    API_GUARD_END
}

int cgo_check_lss_config(const cgo_lss_config *ls) {
    API_GUARD_BEGIN
    std::string why;
    if (int rc = check_lss_config(ls, why)) { set_error(why); return rc; }
    return CGO_OK;
    API_GUARD_END
}

int64_t cgo_lss_default_max_iters(double rho) { return lss_default_max_iters(rho); }

int cgo_solver_destroy(cgo_solver *s) {
    API_GUARD_BEGIN
    if (s) {
        (void)hipSetDevice(s->ctx->c.device);
        cgo_objective *o = s->obj;
        delete s;
        obj_unref(o);
    }
    return CGO_OK;
    API_GUARD_END
}

int cgo_solver_set_x0_host(cgo_solver *s, const double *x0) {
    API_GUARD_BEGIN
    REQUIRE(s && x0, "null argument");
    return s->be->set_x0_host(x0);
    API_GUARD_END
}

int cgo_solver_set_x0_device(cgo_solver *s, const double *x0_dev) {
    API_GUARD_BEGIN
    REQUIRE(s && x0_dev, "null argument");
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, x0_dev) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != s->ctx->c.device) {
        (void)hipGetLastError();
        set_error("cgo_solver_set_x0_device: not a device pointer of this context's GPU");
        return CGO_EINVAL;
    }
    return s->be->set_x0_device(x0_dev);
    API_GUARD_END
}

int cgo_solver_results_device(cgo_solver *s, double *minimizer_dev, double *gradient_dev) {
    API_GUARD_BEGIN
    REQUIRE(s, "null argument");
    for (double *p : {minimizer_dev, gradient_dev}) {
        if (!p) continue;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != s->ctx->c.device) {
            (void)hipGetLastError();
            set_error("cgo_solver_results_device: not a device pointer of this context's GPU");
            return CGO_EINVAL;
        }
    }
    return s->be->download_device(minimizer_dev, gradient_dev);
    API_GUARD_END
}

int cgo_solver_set_x0_fill(cgo_solver *s, int32_t kind, uint64_t seed, double lo, double hi) {
    API_GUARD_BEGIN
    REQUIRE(s, "null argument");
    REQUIRE(kind >= 0 && kind <= 2, "unknown fill kind");
    return s->be->set_x0_fill(kind, seed, lo, hi);
    API_GUARD_END
}

int cgo_solver_start(cgo_solver *s) {
    API_GUARD_BEGIN
    REQUIRE(s, "null argument");
    return s->sv->start();
    API_GUARD_END
}

int cgo_solver_iterate(cgo_solver *s, int64_t iters, int32_t *finished) {
    API_GUARD_BEGIN
    REQUIRE(s, "null argument");
    bool fin = false;
    int rc = s->sv->iterate(iters, fin);
    if (rc == CGO_ESTATE) set_error("cgo_solver_iterate before cgo_solver_start");
    if (finished) *finished = fin ? 1 : 0;
    // A solve that has just reached a terminal status is about to be read: a reduction tail that gave up on a row (NaN in the
    // sums) must surface HERE, not only in cgo_solver_results — primalbarriermethod!, rerun chains and
    // cgo_solver_results_device consume the status first.
    if (rc == CGO_OK && fin) rc = s->be->tail_errors();
    return rc;
    API_GUARD_END
}

int cgo_solver_results(cgo_solver *s, cgo_results *out) {
    API_GUARD_BEGIN
    REQUIRE(s && out, "null argument");
    Solver &sv = *s->sv;
    out->objective = sv.objective();
    out->iters_ran = sv.finished() ? sv.iters_ran() : (int64_t)sv.trace_objective().size();
    out->status = sv.status();
    out->total_fdf_evals = sv.total_evals();
    out->total_launches = s->be->launches();
    if (!sv.finished() && !sv.config().trace_enabled) out->iters_ran = -1;
    const size_t k = sv.trace_objective().size();
    if (out->trace_objective && k) std::memcpy(out->trace_objective, sv.trace_objective().data(), k * sizeof(double));
    if (out->trace_grad_norm && k) std::memcpy(out->trace_grad_norm, sv.trace_grad_norm().data(), k * sizeof(double));
    if (out->trace_step_size && k) std::memcpy(out->trace_step_size, sv.trace_step_size().data(), k * sizeof(double));
    if (out->trace_objective_evals && k) std::memcpy(out->trace_objective_evals, sv.trace_evals().data(), k * sizeof(int64_t));
    if (int rc = s->be->tail_errors()) return rc;
    if (out->minimizer || out->gradient) return s->be->download(out->minimizer, out->gradient);
    return CGO_OK;
    API_GUARD_END
}

int cgo_solver_trial_log(cgo_solver *s, int64_t cap, double *a, double *phi, double *dphi, int64_t *count) {
    API_GUARD_BEGIN
    REQUIRE(s && count, "null argument");
    if (cap < 0) { s->sv->set_log_enabled(true); *count = 0; return CGO_OK; }  // cap < 0: switch the log on
    const auto &L = s->sv->trial_log();
    *count = (int64_t)L.size();
    const int64_t m = std::min<int64_t>(cap, (int64_t)L.size());
    for (int64_t i = 0; i < m; ++i) {
        if (a) a[i] = L[i].a;
        if (phi) phi[i] = L[i].phi;
        if (dphi) dphi[i] = L[i].dphi;
    }
    return CGO_OK;
    API_GUARD_END
}

const char *cgo_solver_kernel_family(cgo_solver *s) {
    if (!s) return "";
    const bool qn = s->sv->config().beta.kind == CGO_BETA_LBFGS;
    if (s->obj->o.two_phase()) return qn ? "k_lse (two-phase) + k_lbfgs" : "k_lse (two-phase)";
    if (s->be->rmode()) {
        const int mp = s->be->max_points();
        return mp >= 7 ? "k_cg (gradient-free, 7-point)" : mp >= 5 ? "k_cg (gradient-free, 5-point)" : (mp >= 3 ? "k_cg (gradient-free, 3-point)" : "k_cg (gradient-free, 1-point)");
    }
    return qn ? "k_fused (stored gradient) + k_lbfgs" : "k_fused (stored gradient)";
}

int64_t cgo_solver_controller_launches(cgo_solver *s) { return s ? s->be->ctl_served() : 0; }

int cgo_solver_resident_stats(cgo_solver *s, int64_t *slices, int64_t *iterations, int64_t *gave_up) {
    API_GUARD_BEGIN
    REQUIRE(s, "null argument");
    if (slices) *slices = s->be->resident_slices();
    if (iterations) *iterations = s->be->resident_iters();
    if (gave_up) *gave_up = s->be->resident_gave_up();
    return CGO_OK;
    API_GUARD_END
}

int cgo_solver_lbfgs_stats(cgo_solver *s, int64_t *speculated, int64_t *fused, int64_t *plain) {
    API_GUARD_BEGIN
    REQUIRE(s, "null argument");
    if (speculated) *speculated = s->be->push_count(0);
    if (fused) *fused = s->be->push_count(1);
    if (plain) *plain = s->be->push_count(2);
    return CGO_OK;
    API_GUARD_END
}

int cgo_solver_kernel_symbol(cgo_solver *s, int32_t kernel_kind, char *buf, int32_t cap) {
    API_GUARD_BEGIN
    REQUIRE(s && buf && cap > 0, "bad argument");
    const std::string sym = s->be->kernel_symbol(kernel_kind);
    std::snprintf(buf, (size_t)cap, "%s", sym.c_str());
    return CGO_OK;
    API_GUARD_END
}

int cgo_solver_profile_enable(cgo_solver *s, int32_t on) {
    API_GUARD_BEGIN
    REQUIRE(s, "null argument");
    s->be->profile_enable(on != 0);
    return CGO_OK;
    API_GUARD_END
}
int cgo_solver_profile_reset(cgo_solver *s) {
    API_GUARD_BEGIN
    REQUIRE(s, "null argument");
    s->be->profile_reset();
    return CGO_OK;
    API_GUARD_END
}
int cgo_solver_profile_get(cgo_solver *s, int32_t kind, int64_t *launches, double *ms, double *bytes) {
    API_GUARD_BEGIN
    REQUIRE(s && launches && ms && bytes, "null argument");
    s->be->profile_get(kind, launches, ms, bytes);
    return CGO_OK;
    API_GUARD_END
}

// ---- one-shot drop-ins ----------------------------------------------------------
int cgo_minimize(cgo_ctx *ctx, cgo_objective *obj, const double *x0, const cgo_cg_config *cfg,
                 const cgo_ls_config *ls, cgo_results *out) {
    API_GUARD_BEGIN
    REQUIRE(x0 && out, "null argument");
    cgo_solver *s = nullptr;
    int rc = cgo_solver_create(ctx, obj, cfg, ls, &s);
    if (rc) return rc;
    rc = s->be->set_x0_host(x0);
    if (!rc) rc = s->sv->start();
    bool fin = false;
    while (!rc && !fin) rc = s->sv->iterate(INT64_MAX / 2, fin);
    if (!rc) rc = cgo_solver_results(s, out);
    std::string keep = get_error();
    cgo_solver_destroy(s);
    if (rc) set_error(keep);
    return rc;
    API_GUARD_END
}

int cgo_evalwolfeconditions(const cgo_ls_config *ls, double phi_a, double dphi_a, double a, double uu,
                            double phi_0, double dphi_0, int32_t *valid_large, int32_t *valid_small) {
    API_GUARD_BEGIN
    REQUIRE(ls && valid_large && valid_small, "null argument");
    REQUIRE(ls->cond_kind == CGO_COND_WOLFE || ls->cond_kind == CGO_COND_YUAN_WEI_LU, "not a Wolfe-type condition");
    if (ls->cond_kind == CGO_COND_WOLFE) {
        if (!(0.0 < ls->c1 && ls->c1 < ls->c2 && ls->c2 < 1.0)) { set_error("AssertionError: zero(T) < c1 < c2 < one(T)  (wolfe.jl:278)"); return CGO_EINVAL; }
    } else if (!(0.0 < ls->delta1 && ls->delta1 < ls->c1 && ls->c1 < ls->c2 && ls->c2 < 1.0)) {
        set_error("AssertionError: zero(T) < δ1 < c1 < c2 < one(T)  (wolfe.jl:233)"); return CGO_EINVAL;
    }
    bool okl = false, oks = false;
    wolfe_tests(*ls, phi_0, dphi_0, uu, phi_a, dphi_a, a, okl, oks);
    *valid_large = okl; *valid_small = oks;
    return CGO_OK;
    API_GUARD_END
}

int cgo_evalbacktrackcondition(const cgo_ls_config *ls, double phi_a, double a, double phi_0, double dphi_0,
                               int32_t *valid) {
    API_GUARD_BEGIN
    REQUIRE(ls && valid, "null argument");
    if (!(0.0 < ls->c1 && ls->c1 < 1.0)) { set_error("AssertionError: zero(T) < c1 < one(T)  (geometric.jl:169)"); return CGO_EINVAL; }
    *valid = armijo_test(ls->c1, phi_a, a, phi_0, dphi_0);
    return CGO_OK;
    API_GUARD_END
}

int cgo_solvesystem(cgo_ctx *ctx, cgo_objective *obj, const double *x0, const cgo_cg_config *cfg,
                    const cgo_lss_config *ls, cgo_results *out) {
    API_GUARD_BEGIN
    REQUIRE(x0 && out, "null argument");
    cgo_solver *s = nullptr;
    int rc = cgo_solver_create_sys(ctx, obj, cfg, ls, &s);
    if (rc) return rc;
    rc = s->be->set_x0_host(x0);
    if (!rc) rc = s->sv->start();
    bool fin = false;
    while (!rc && !fin) rc = s->sv->iterate(INT64_MAX / 2, fin);
    if (!rc) rc = cgo_solver_results(s, out);
    std::string keep = get_error();
    cgo_solver_destroy(s);
    if (rc) set_error(keep);
    return rc;
    API_GUARD_END
}

// One stage of a rerun chain: x_initial from the host (first stage) or from a device buffer (the previous stage's minimizer:
// no PCIe round trip between stages), results to the caller's host buffers where it supplied them, and the minimizer to
// `seed_out` (device) if another stage may follow.
static int rerun_stage(cgo_ctx *ctx, cgo_objective *obj, const double *x0_host, const double *x0_dev, const cgo_cg_config *cfg,
                       const cgo_ls_config *ls, cgo_results *out, double *seed_out) {
    cgo_solver *s = nullptr;
    int rc = cgo_solver_create(ctx, obj, cfg, ls, &s);
    if (rc) return rc;
    rc = x0_dev ? s->be->set_x0_device(x0_dev) : s->be->set_x0_host(x0_host);
    if (!rc) rc = s->sv->start();
    bool fin = false;
    while (!rc && !fin) rc = s->sv->iterate(INT64_MAX / 2, fin);
    if (!rc) rc = cgo_solver_results(s, out);
    if (!rc && seed_out && out->status != CGO_SUCCESS) rc = s->be->download_device(seed_out, nullptr);
    std::string keep = get_error();
    cgo_solver_destroy(s);
    if (rc) set_error(keep);
    return rc;
}

int cgo_minimize_rerun(cgo_ctx *ctx, cgo_objective *obj, const double *x0, const cgo_cg_config *cfg,
                       const cgo_ls_config *ls, const cgo_cg_config *rerun_cfgs,
                       const cgo_ls_config *rerun_ls, int32_t npairs, cgo_results *outs, int32_t *nouts) {
    API_GUARD_BEGIN
    REQUIRE(ctx && obj && x0 && outs && nouts && npairs >= 0, "bad argument");
    REQUIRE(npairs == 0 || (rerun_cfgs && rerun_ls), "null rerun configs");
    // Each stage restarts from the previous stage's minimizer (optim.jl:195-200).  That vector never leaves the GPU: it is
    // copied device to device into `seed` (0.3 ms at n = 1e8 instead of 0.8 GB over PCIe each way); host copies of
    // minimizer / gradient are made only into the buffers the caller supplied (either may be NULL for any stage).  On a
    // sharded context every rank runs the same chain on its own shard — the status that decides whether another stage
    // follows is formed from the merged scalars and therefore identical on all ranks.
    DevBuf seed;
    if (npairs > 0) {
        if (hipSetDevice(ctx->c.device) != hipSuccess) { set_error("hipSetDevice failed"); return CGO_EHIP; }
        if (int rc = seed.alloc((size_t)obj->o.n_local)) return rc;
    }
    int rc = rerun_stage(ctx, obj, x0, nullptr, cfg, ls, &outs[0], npairs > 0 ? seed.p : nullptr);  // optim.jl:183-188
    if (rc) return rc;
    int cnt = 1;
    for (int k = 0; k < npairs; ++k) {                           // optim.jl:191-205
        if (outs[cnt - 1].status == CGO_SUCCESS) break;
        rc = rerun_stage(ctx, obj, nullptr, seed.p, &rerun_cfgs[k], &rerun_ls[k], &outs[cnt], k + 1 < npairs ? seed.p : nullptr);
        if (rc) return rc;
        cnt++;
    }
    *nouts = cnt;
    return CGO_OK;
    API_GUARD_END
}

// ---- kernel-level entry points --------------------------------------------------
int cgo_kernel_dir(cgo_ctx *ctx, double *u, const double *g, double beta, int64_t n, double *out2) {
    API_GUARD_BEGIN
    REQUIRE(ctx && u && g && out2 && n >= 1, "bad argument");
    return HipBackend::run_dir(&ctx->c, u, g, beta, n, out2);
    API_GUARD_END
}

int cgo_kernel_beta_partials(cgo_ctx *ctx, const double *gn, const double *g, const double *u,
                             int64_t n, double *out9) {
    API_GUARD_BEGIN
    REQUIRE(ctx && gn && g && u && out9 && n >= 1, "bad argument");
    return HipBackend::run_beta_partials(&ctx->c, gn, g, u, n, out9);
    API_GUARD_END
}

int cgo_getbeta(cgo_ctx *ctx, const cgo_beta_config *b, const double *gn, const double *g,
                const double *u, int64_t n, double *beta) {
    API_GUARD_BEGIN
    REQUIRE(ctx && b && gn && g && u && beta && n >= 1, "bad argument");
    REQUIRE(b->kind >= 0 && b->kind < CGO_BETA_LBFGS, "getβ is defined for the CGβConfig kinds");
    double p[9];
    int rc = HipBackend::run_beta_partials(&ctx->c, gn, g, u, n, p);
    if (rc) return rc;
    TrialSums t;
    t.f = 0.0; t.gtu = p[0]; t.gtgt = p[1]; t.gtg = p[2]; t.yy = p[3]; t.uy = p[4]; t.ygt = p[5];
    const double gg = p[6], gu_old = p[7], uu = p[8];
    BetaNorms bn = beta_norms_fast(t, uu);
    if (!beta_norms_fast_ok(b->kind, t, uu)) {   // LinearAlgebra.norm's scaled form, from the host operands at hand
        auto nrm2 = [n](const double *v, const double *w) {
            double m = 0.0;
            for (int64_t i = 0; i < n; ++i) { const double a = std::fabs(w ? v[i] - w[i] : v[i]); if (a != a) return a; if (a > m) m = a; }
            if (m == 0.0 || std::isinf(m)) return m;
            double ss = 0.0;
            for (int64_t i = 0; i < n; ++i) { const double r = (w ? v[i] - w[i] : v[i]) / m; ss += r * r; }
            return m * std::sqrt(ss);
        };
        if (!sumsq_in_range(uu)) bn.u = nrm2(u, nullptr);
        if (!sumsq_in_range(t.yy)) bn.y = nrm2(gn, g);
        if (!sumsq_in_range(t.gtgt)) bn.gt = nrm2(gn, nullptr);
    }
    *beta = beta_from_sums(b->kind, b->mu, t, gu_old, gg, uu, bn);
    return CGO_OK;
    API_GUARD_END
}

int cgo_kernel_trial(cgo_objective *obj, const double *x, const double *u, double a,
                     double *g_next_out, double *out2) {
    API_GUARD_BEGIN
    REQUIRE(obj && x && u && out2, "bad argument");
    REQUIRE(!obj->o.host_closure() && obj->o.kind != CGO_OBJ_ROSENBROCK_CHAINED,
            "cgo_kernel_trial is an entry point of the element-wise kernel family: not defined for a host closure or the stencil objective");
    return HipBackend::run_trial(&obj->o, x, u, a, g_next_out, out2);
    API_GUARD_END
}

int cgo_solver_placement_info(cgo_solver *s, double *as_allocated_us, double *chosen_us, int32_t *candidates) {
    API_GUARD_BEGIN
    REQUIRE(s && as_allocated_us && chosen_us && candidates, "null argument");
    int c = 0;
    s->be->placement_info(as_allocated_us, chosen_us, &c);
    *candidates = c;
    return CGO_OK;
    API_GUARD_END
}

int cgo_bench_stream_mix(cgo_ctx *ctx, int64_t n, int32_t reps, double *median_us, double *best_us) {
    API_GUARD_BEGIN
    REQUIRE(ctx && median_us && best_us, "bad argument");
    return HipBackend::bench_stream_mix(&ctx->c, n, reps, median_us, best_us);
    API_GUARD_END
}

int cgo_bench_kernel(cgo_ctx *ctx, cgo_objective *obj, int32_t kernel_kind, int64_t n, int32_t reps,
                     double *ms, double *bytes) {
    API_GUARD_BEGIN
    REQUIRE(ctx && ms && bytes, "bad argument");
    return HipBackend::bench_kernel(&ctx->c, obj ? &obj->o : nullptr, kernel_kind, n, reps, ms, bytes);
    API_GUARD_END
}

}  // extern "C"
