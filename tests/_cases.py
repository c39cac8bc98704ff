"""Shared test plumbing: one `Case` description, four runners.

    run_oracle   oracle/cgo_oracle.c      (the checker)
    run_numpy    oracle/cgo_oracle_np.py  (independent second restatement)
    run_hostsim  product host engine over the test-double backend (CPU tier)
    run_gpu      product: libcgo_hip.so through the C ABI (GPU tier)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from dataclasses import dataclass, field

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from oracle import cgo_oracle_np as N  # noqa: E402

SEED = 24  # examples/min.jl:7


@dataclass
class Case:
    name: str
    objective: str            # booth | quad_diag | rosenbrock_paired
    n: int
    x0: np.ndarray
    beta: str = "HagerZhang"  # oracle.BETA_KINDS key
    mu: float = 0.1
    m: int = 10
    ls: str = "StrongWolfeBisection"
    c1: float = 1e-5
    c2: float = 0.8
    growth: float = 2.0
    ls_max_iters: int = 1000
    zoom_max_iters: int = 100
    cond: str = "Wolfe"
    delta1: float = 1e-4
    max_step_size: float = 1e12
    feas_max_iters: int = 50
    discount: float = 0.5
    sys_s: float = 1.0          # ls == "SolveSys": LinesearchSolveSys(ρ, σ, s, max_iters)  (solve_system.jl:6-27)
    sys_sigma: float = 0.5
    sys_rho: float = 0.95
    sys_max_iters: int | None = None  # None = the reference's default round(Int, log(ρ, 1e-6))
    eps: float = 1e-5
    max_iters: int = 1000
    trace: bool = True
    D: np.ndarray | None = None
    lam: float = 0.0
    tol: float = 1e-10
    extra: dict = field(default_factory=dict)


@dataclass
class Out:
    objective: float
    minimizer: np.ndarray
    gradient: np.ndarray
    iters_ran: int
    status: str
    trace_objective: np.ndarray
    trace_grad_norm: np.ndarray
    trace_step_size: np.ndarray
    trace_objective_evals: np.ndarray
    log_a: np.ndarray
    log_phi: np.ndarray
    log_dphi: np.ndarray
    total_fdf_evals: int = 0
    total_launches: int = 0
    controller_launches: int = 0  # GPU only: launches armed by the on-device controller
    lbfgs_pushes: tuple = (0, 0, 0)  # GPU only: L-BFGS state updates (speculated, fused, plain) — cgo_solver_lbfgs_stats


def quad_D(n, lo=1.0, hi=1000.0, seed=SEED):
    return O.fill_uniform(n, seed, lo, hi)


# ------------------------------------------------------------------ oracle (C)
def _orc_ls(c: Case):
    if c.ls == "SolveSys":
        return O.linesearch_solve_sys(c.sys_s, c.sys_sigma, c.sys_rho, c.sys_max_iters)
    if c.ls == "StrongWolfeBisection":
        return O.strong_wolfe(c.c1, c.c2, c.growth, c.ls_max_iters, c.zoom_max_iters)
    if c.ls == "Backtracking":
        return O.backtracking(c.c1, c.discount, c.ls_max_iters, c.feas_max_iters)
    return O.wolfe_bisection(c.cond, c.c1, c.c2, c.delta1, c.ls_max_iters, c.max_step_size,
                             c.feas_max_iters)


def run_oracle(c: Case) -> Out:
    obj = O.python_objective(c.extra["fdf"]) if c.objective == "closure" else O.objective(c.objective, D=c.D, lam=c.lam)
    cfg = O.cg_config(c.eps, O.beta_config(c.beta, c.mu, c.m), c.max_iters, c.trace)
    run = O.solvesystem if c.ls == "SolveSys" else O.minimizeobjective
    r = run(obj, c.x0, cfg, _orc_ls(c), log_cap=200000)
    return Out(r.objective, r.minimizer, r.gradient, r.iters_ran, r.status, r.trace_objective,
               r.trace_grad_norm, r.trace_step_size, r.trace_objective_evals, r.log_a, r.log_phi,
               r.log_dphi, r.total_fdf_evals)


# ------------------------------------------------------------------ numpy oracle
def run_numpy(c: Case) -> Out:
    fdf = {"booth": N.booth, "rosenbrock_paired": N.rosenbrock_paired,
           "rosenbrock_chained": N.rosenbrock_chained}.get(c.objective)
    if c.objective == "closure":   # the reference's own contract: f = fdf!(g, x)
        fdf = c.extra["fdf"]
    if c.objective == "quad_diag":
        fdf = N.make_quad_diag(c.D)
    if c.objective == "lse":
        fdf = N.make_lse(c.lam)
    beta = {"HagerZhang": N.HagerZhang(), "YuanWangSheng": N.YuanWangSheng(c.mu),
            "SallehAlhawarat": N.SallehAlhawarat(), "LiuStorrey": N.LiuStorrey(),
            "PolakRibiere": N.PolakRibiere(), "HestenesStiefel": N.HestenesStiefel(),
            "DaiYuan": N.DaiYuan(), "LBFGS": N.LBFGS(c.m), "BroydenFamily": N.BroydenFamily(c.mu)}[c.beta]
    if c.ls == "SolveSys":
        r = N.solvesystem(fdf, c.x0, N.CGConfig(c.eps, beta, c.max_iters, c.trace),
                          N.LinesearchSolveSys(c.sys_s, c.sys_sigma, c.sys_rho, c.sys_max_iters))
        lg = np.array(r.log, dtype=np.float64).reshape(-1, 3)
        return Out(r.objective, r.minimizer, r.gradient, r.iters_ran, r.status,
                   np.array(r.trace_objective), np.array(r.trace_grad_norm),
                   np.array(r.trace_step_size), np.array(r.trace_objective_evals, dtype=np.int64),
                   lg[:, 0], lg[:, 1], lg[:, 2])
    if c.ls == "StrongWolfeBisection":
        ls = N.StrongWolfeBisection(c.c1, c.c2, c.growth, c.ls_max_iters, c.zoom_max_iters)
    elif c.ls == "Backtracking":
        ls = N.Backtracking(N.Armijo(c.c1), c.discount, c.ls_max_iters, c.feas_max_iters)
    else:
        cond = N.Wolfe(c.c1, c.c2) if c.cond == "Wolfe" else N.YuanWeiLuWolfe(c.c1, c.c2, c.delta1)
        ls = N.WolfeBisection(cond, c.ls_max_iters, c.max_step_size, c.feas_max_iters)
    r = N.minimizeobjective(fdf, c.x0, N.CGConfig(c.eps, beta, c.max_iters, c.trace), ls)
    lg = np.array(r.log, dtype=np.float64).reshape(-1, 3)
    return Out(r.objective, r.minimizer, r.gradient, r.iters_ran, r.status,
               np.array(r.trace_objective), np.array(r.trace_grad_norm),
               np.array(r.trace_step_size), np.array(r.trace_objective_evals, dtype=np.int64),
               lg[:, 0], lg[:, 1], lg[:, 2])


# ------------------------------------------------------------------ product structs (shared by hostsim & gpu)
def _product_structs(c: Case):
    import cgo_amd as cgo
    from cgo_amd import _lib
    beta = {"HagerZhang": cgo.HagerZhang(), "YuanWangSheng": cgo.YuanWangSheng(c.mu),
            "SallehAlhawarat": cgo.SallehAlhawarat(), "LiuStorrey": cgo.LiuStorrey(),
            "PolakRibiere": cgo.PolakRibiere(), "HestenesStiefel": cgo.HestenesStiefel(),
            "DaiYuan": cgo.DaiYuan(), "LBFGS": cgo.LBFGS(c.m), "BroydenFamily": cgo.BroydenFamily(c.mu)}[c.beta]
    cfg = cgo.CGConfig(c.eps, beta, c.max_iters, False,
                       cgo.EnableTrace() if c.trace else cgo.DisableTrace())
    if c.ls == "SolveSys":
        ls = cgo.setupLinesearchSolveSys(c.sys_s, σ=c.sys_sigma, ρ=c.sys_rho, max_iters=c.sys_max_iters)
    elif c.ls == "StrongWolfeBisection":
        ls = cgo.StrongWolfeBisection(c.c1, c.c2, c.growth, c.ls_max_iters, c.zoom_max_iters)
    elif c.ls == "Backtracking":
        ls = cgo.Backtracking(cgo.Armijo(c.c1), c.discount, c.ls_max_iters, c.feas_max_iters)
    else:
        cond = cgo.Wolfe(c.c1, c.c2) if c.cond == "Wolfe" else cgo.YuanWeiLuWolfe(c.c1, c.c2, c.delta1)
        ls = cgo.WolfeBisection(cond, c.ls_max_iters, c.max_step_size, c.feas_max_iters)
    return cgo, _lib, cfg, ls


# ------------------------------------------------------------------ hostsim
_SIM = None
_SIM_DIR = os.path.join(ROOT, "tests", "hostsim")


def sim_lib():
    global _SIM
    if _SIM is None:
        subprocess.run(["make", "-C", _SIM_DIR, "-s"], check=True)
        from cgo_amd import _lib
        L = C.CDLL(os.path.join(_SIM_DIR, "_build", "libcgo_hostsim.so"))
        dp, i64p = _lib.dp, _lib.i64p
        L.sim_minimize.restype = C.c_int
        L.sim_minimize.argtypes = [C.c_int, C.c_int64, C.c_int64, dp, C.c_double, dp,
                                   C.POINTER(_lib.CGConfigC), C.POINTER(_lib.LSConfigC), C.c_int,
                                   C.c_int, _lib.ALLGATHER_FN, C.c_void_p, C.c_int64,
                                   C.POINTER(_lib.ResultsC), C.c_int64, dp, dp, dp, i64p]
        L.sim_solvesystem.restype = C.c_int
        L.sim_solvesystem.argtypes = [C.c_int, C.c_int64, C.c_int64, dp, C.c_double, dp,
                                      C.POINTER(_lib.CGConfigC), C.POINTER(_lib.LSSConfigC), C.c_int,
                                      C.c_int, _lib.ALLGATHER_FN, C.c_void_p, C.c_int64,
                                      C.POINTER(_lib.ResultsC), C.c_int64, dp, dp, dp, i64p]
        L.sim_set_host_objective.restype = None
        L.sim_set_host_objective.argtypes = [_lib.FDF_FN, C.c_void_p]
        L.sim_set_points.restype = None
        L.sim_set_points.argtypes = [C.c_int]
        L.sim_set_ctl_depth.restype = None
        L.sim_set_ctl_depth.argtypes = [C.c_int]
        L.sim_ctl_stats.restype = None
        L.sim_ctl_stats.argtypes = [i64p, i64p]
        L.sim_set_lbfgs_spec.restype = None
        L.sim_set_lbfgs_spec.argtypes = [C.c_int]
        L.sim_lbfgs_spec_stats.restype = None
        L.sim_lbfgs_spec_stats.argtypes = [i64p, i64p, i64p]
        L.sim_set_fuse_grad.restype = None
        L.sim_set_fuse_grad.argtypes = [C.c_int]
        L.sim_fused_pushes.restype = C.c_int64
        L.sim_fused_pushes.argtypes = []
        L.sim_set_resident.restype = None
        L.sim_set_resident.argtypes = [C.c_int, C.c_int64]
        L.sim_resident_stats.restype = None
        L.sim_resident_stats.argtypes = [i64p, i64p, i64p]
        L.sim_beta_from_scalars.restype = C.c_double
        L.sim_beta_from_scalars.argtypes = [C.POINTER(_lib.BetaConfig), dp, C.c_double, C.c_double,
                                            C.c_double]
        _SIM = L
    return _SIM


def run_hostsim(c: Case, rank=0, world=1, allgather=None, chunk=0, ctl_depth=0, ctl_stats=None, points=3,
                resident=False, resident_log_cap=0, resident_stats=None, lbfgs_spec=0, lbfgs_spec_stats=None, fuse_grad=True) -> Out:
    """ctl_depth > 0 switches on the emulated on-device controller (csrc/cgo_ctl.hpp);
    ctl_stats (a dict) receives how many rounds it ran and how many launches it served.
    resident=True runs whole iterations through res_iterate (csrc/cgo_resident.hpp), the loop every thread of the
    resident kernel runs; resident_stats receives slices / iterations inside slices / hand-backs to the host."""
    cgo, _lib, cfg, ls = _product_structs(c)
    L = sim_lib()
    L.sim_set_ctl_depth(int(ctl_depth))
    L.sim_set_points(int(points))
    L.sim_set_lbfgs_spec(int(lbfgs_spec))   # the one-ring-pass L-BFGS protocol of the product backend (0 off, 1 own state-update launch, 2 deferred)
    L.sim_set_fuse_grad(1 if fuse_grad else 0)
    L.sim_set_resident(1 if resident else 0, int(resident_log_cap))
    dp, i64p = _lib.dp, _lib.i64p
    off, nloc = cgo.shard_extent(c.n, rank, world)
    x0 = np.ascontiguousarray(c.x0[off:off + nloc], dtype=np.float64)
    p0 = np.ascontiguousarray(c.D[off:off + nloc]) if c.D is not None else None
    cap = max(c.max_iters, 1)
    x, g = np.empty(nloc), np.empty(nloc)
    to, tg, ts, te = np.zeros(cap), np.zeros(cap), np.zeros(cap), np.zeros(cap, dtype=np.int64)
    r = _lib.ResultsC()
    r.minimizer, r.gradient = x.ctypes.data_as(dp), g.ctypes.data_as(dp)
    r.trace_objective, r.trace_grad_norm = to.ctypes.data_as(dp), tg.ctypes.data_as(dp)
    r.trace_step_size, r.trace_objective_evals = ts.ctypes.data_as(dp), te.ctypes.data_as(i64p)
    LC = 200000
    la, lp, ld = np.zeros(LC), np.zeros(LC), np.zeros(LC)
    ll = C.c_int64(0)

    def tramp(_u, send, recv, count):
        s = np.ctypeslib.as_array(send, shape=(count,)).copy()
        out = np.asarray(allgather(s), dtype=np.float64).reshape(-1)
        np.ctypeslib.as_array(recv, shape=(world * count,))[:] = out
        return 0
    cb = _lib.ALLGATHER_FN(tramp) if allgather else _lib.ALLGATHER_FN(0)
    cc, lc = cfg._c(), ls._c()
    kind = {"quad_diag": 0, "rosenbrock_paired": 1, "booth": 2, "lse": 3, "closure": 5}[c.objective]   # (lse: single rank in the test double)
    if kind == 5:
        pyfdf = c.extra["fdf"]

        def ftramp(_u, gp, xp, n_):
            return float(pyfdf(np.ctypeslib.as_array(gp, shape=(n_,)), np.ctypeslib.as_array(xp, shape=(n_,))))
        fcb = _lib.FDF_FN(ftramp)
        L.sim_set_host_objective(fcb, None)
    entry = L.sim_solvesystem if c.ls == "SolveSys" else L.sim_minimize
    rc = entry(kind, nloc, off, p0.ctypes.data_as(dp) if p0 is not None else None, c.lam,
                        x0.ctypes.data_as(dp), C.byref(cc), C.byref(lc), rank, world, cb, None,
                        chunk, C.byref(r), LC, la.ctypes.data_as(dp), lp.ctypes.data_as(dp),
                        ld.ctypes.data_as(dp), C.byref(ll))
    L.sim_set_ctl_depth(0)
    L.sim_set_points(3)
    L.sim_set_lbfgs_spec(0)
    L.sim_set_fuse_grad(1)
    L.sim_set_resident(0, 0)
    assert rc == 0, f"sim_minimize rc={rc}"
    if lbfgs_spec_stats is not None:
        a, b, h = C.c_int64(), C.c_int64(), C.c_int64()
        L.sim_lbfgs_spec_stats(C.byref(a), C.byref(b), C.byref(h))
        lbfgs_spec_stats["pushes"], lbfgs_spec_stats["rode"], lbfgs_spec_stats["flushed"] = a.value, b.value, h.value
        lbfgs_spec_stats["fused"] = int(L.sim_fused_pushes())
    if resident_stats is not None:
        a, b, h = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        L.sim_resident_stats(C.byref(a), C.byref(b), C.byref(h))
        resident_stats["slices"], resident_stats["iters"], resident_stats["host"] = a.value, b.value, h.value
    if ctl_stats is not None:
        a, b = C.c_int64(0), C.c_int64(0)
        L.sim_ctl_stats(C.byref(a), C.byref(b))
        ctl_stats["rounds"], ctl_stats["served"] = a.value, b.value
    k = int(r.iters_ran) if c.trace else 0
    n = ll.value
    return Out(r.objective, x, g, int(r.iters_ran), O.STATUS_NAMES[r.status], to[:k], tg[:k],
               ts[:k], te[:k], la[:n], lp[:n], ld[:n], int(r.total_fdf_evals), int(r.total_launches))


# ------------------------------------------------------------------ GPU (product, through the C ABI)
def gpu_objective(c: Case, ctx=None):
    import cgo_amd as cgo
    if c.objective == "quad_diag":
        return cgo.QuadDiag(c.D, ctx)
    if c.objective == "rosenbrock_paired":
        return cgo.RosenbrockPaired(c.n, ctx)
    if c.objective == "booth":
        return cgo.Booth(ctx)
    if c.objective == "lse":
        return cgo.LogSumExp(c.n, c.lam, ctx)
    if c.objective == "rosenbrock_chained":
        return cgo.RosenbrockChained(c.n, ctx)
    if c.objective == "closure":   # host closure through cgo_objective_create_callback: GPU engine, objective on the host
        return cgo.HostObjective(c.extra["fdf"], c.n, ctx)
    raise KeyError(c.objective)


def run_gpu(c: Case, ctx=None, chunk=0) -> Out:
    cgo, _lib, cfg, ls = _product_structs(c)
    obj = gpu_objective(c, ctx)
    s = cgo.Solver(obj, cfg, ls)
    try:
        s.enable_trial_log()
        s.set_x0(c.x0)
        s.start()
        while not s.iterate(chunk if chunk > 0 else 1 << 40):
            pass
        r = s.results()
        la, lp, ld = s.trial_log()
        served = s.controller_launches()
        pushes = s.lbfgs_stats()
    finally:
        s.close()
        obj.close()
    return Out(r.objective, r.minimizer, r.gradient, r.iters_ran, r.status, r.trace.objective,
               r.trace.grad_norm, r.trace.step_size, r.trace.objective_evals, la, lp, ld,
               r.total_fdf_evals, r.total_launches, served, pushes)


BIGN = "9000000000000000000"


def pin_points(monkeypatch, pts: int):
    """Fix the number of trial steps per fused launch (1, 3, 5 or 7) instead of the size/objective policy."""
    m, m5, m7 = {1: (BIGN, BIGN, BIGN), 3: ("0", BIGN, BIGN), 5: ("0", "0", BIGN), 7: ("0", "0", "0")}[pts]
    monkeypatch.setenv("CGO_MULTI_MIN_N", m)
    monkeypatch.setenv("CGO_MULTI5_MIN_N", m5)
    monkeypatch.setenv("CGO_MULTI7_MIN_N", m7)


# ------------------------------------------------------------------ comparison
def rel(a, b):
    """‖a − b‖ / ‖b‖, scaled so that it survives entries near the overflow/underflow thresholds."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    sc = float(np.max(np.abs(b))) if b.size else 0.0
    if not np.isfinite(sc) or sc == 0.0:
        sc = 1.0
    d = np.linalg.norm(a / sc - b / sc)
    return float(d / max(np.linalg.norm(b / sc), 1e-300))


def relf(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def first_divergence(a: Out, b: Out, step_rtol=0.0):
    """Index of the first evalϕdϕ! whose step differs (None = same step sequence).

    step_rtol = 0 demands bitwise equal steps — true for the two bisection line searches, whose
    steps are dyadic functions of constants.  Backtracking's first step is |ϕ₀|/u·u
    (geometric.jl:51), i.e. itself a reduction, so there it can only match to rounding."""
    m = min(len(a.log_a), len(b.log_a))
    for i in range(m):
        if a.log_a[i] != b.log_a[i] and not abs(a.log_a[i] - b.log_a[i]) <= step_rtol * abs(b.log_a[i]):
            return i
    return None if len(a.log_a) == len(b.log_a) else m


def assert_parity(got: Out, ref: Out, tol=1e-10, name="", step_rtol=0.0):
    """The north-star bar: same step sequence ⇒ ≤ tol relative on iterate and objective."""
    div = first_divergence(got, ref, step_rtol)
    assert div is None, (f"{name}: step sequence diverges at trial #{div}: "
                         f"a={got.log_a[div] if div < len(got.log_a) else None} vs "
                         f"{ref.log_a[div] if div < len(ref.log_a) else None}")
    assert got.status == ref.status, f"{name}: status {got.status} vs {ref.status}"
    assert got.iters_ran == ref.iters_ran, f"{name}: iters {got.iters_ran} vs {ref.iters_ran}"
    assert np.array_equal(got.trace_objective_evals, ref.trace_objective_evals), f"{name}: evals/iter differ"
    assert np.allclose(got.trace_step_size, ref.trace_step_size, rtol=step_rtol, atol=0), f"{name}: accepted steps differ"
    rx, rf = rel(got.minimizer, ref.minimizer), relf(got.objective, ref.objective)
    assert rx <= tol, f"{name}: minimizer rel diff {rx:.3e} > {tol:g}"
    assert rf <= tol or abs(got.objective - ref.objective) <= 1e-290, f"{name}: objective rel diff {rf:.3e} > {tol:g}"
    if len(ref.trace_objective):
        assert np.allclose(got.trace_objective, ref.trace_objective, rtol=max(tol, 1e-12) * 10, atol=1e-300), \
            f"{name}: objective trace differs"
