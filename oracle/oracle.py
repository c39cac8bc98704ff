"""ctypes loader for the CPU oracle (oracle/cgo_oracle.c).

TEST INFRASTRUCTURE ONLY — may be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg, never by the product package.  See the header
of oracle/cgo_oracle.h for what the oracle restates and how it is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcgo_oracle.so")
_SO_OMP = os.path.join(_HERE, "_build", "libcgo_oracle_omp.so")  # all-cores CPU baseline only
_SO_EXACT = os.path.join(_HERE, "_build", "libcgo_oracle_exact.so")  # the arbiter: reductions in twice the working precision

STATUS_NAMES = [
    "incomplete", "success", "increasing_objective",
    "non_finite_objective_or_gradient_proposed", "max_iters_reached",
    "non_descent_search_direction", "linesearch_a_max_overflow",
    "linesearch_max_iters_reached", "zoom_max_iters_reached",
    "accepted_non_finite_iterate", "cannot_find_initial_feasible_step",
    "max_step_length_reached", "cannot_find_feasible_step",
    "step_bracket_precision_issue", "bisection_lower_bound_larger_than_proposed_step",
    "feasible", "infeasible", "non_finite_step_proposed",
    "proposed_step_same_as_current_step", "linesearch_failed",
]

BETA_KINDS = {
    "HagerZhang": 0, "YuanWangSheng": 1, "SallehAlhawarat": 2, "LiuStorrey": 3,
    "PolakRibiere": 4, "HestenesStiefel": 5, "DaiYuan": 6, "LBFGS": 7, "BroydenFamily": 8,
}
LS_KINDS = {"StrongWolfeBisection": 0, "WolfeBisection": 1, "Backtracking": 2}
COND_KINDS = {"Wolfe": 0, "YuanWeiLuWolfe": 1, "Armijo": 2}

FDF_T = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)


class BetaConfig(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lbfgs_m", C.c_int32), ("mu", C.c_double)]


class LSConfig(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("cond_kind", C.c_int32),
        ("c1", C.c_double), ("c2", C.c_double),
        ("a_max_growth_factor", C.c_double), ("delta1", C.c_double),
        ("max_step_size", C.c_double),
        ("max_iters", C.c_int64), ("zoom_max_iters", C.c_int64),
        ("feasibility_max_iters", C.c_int64), ("discount_factor", C.c_double),
    ]


class CGConfig(C.Structure):
    _fields_ = [
        ("eps", C.c_double), ("beta", BetaConfig), ("max_iters", C.c_int64),
        ("verbose", C.c_int32), ("trace_enabled", C.c_int32),
    ]


class Results(C.Structure):
    _fields_ = [
        ("objective", C.c_double),
        ("minimizer", C.POINTER(C.c_double)), ("gradient", C.POINTER(C.c_double)),
        ("iters_ran", C.c_int64), ("status", C.c_int32), ("_pad", C.c_int32),
        ("trace_objective", C.POINTER(C.c_double)), ("trace_grad_norm", C.POINTER(C.c_double)),
        ("trace_step_size", C.POINTER(C.c_double)), ("trace_objective_evals", C.POINTER(C.c_int64)),
        ("log_cap", C.c_int64), ("log_len", C.c_int64),
        ("log_a", C.POINTER(C.c_double)), ("log_phi", C.POINTER(C.c_double)),
        ("log_dphi", C.POINTER(C.c_double)),
        ("total_fdf_evals", C.c_int64),
        ("log_margin", C.POINTER(C.c_double)),
        ("nsnap", C.c_int64), ("snap_done", C.c_int64),
        ("snap_iters", C.POINTER(C.c_int64)), ("snap_x", C.POINTER(C.c_double)),
        ("trace_time", C.POINTER(C.c_double)),
    ]


class LSSConfig(C.Structure):  # LinesearchSolveSys (solve_system.jl:6-11)
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("s", C.c_double), ("max_iters", C.c_int64)]


class QuadParams(C.Structure):
    _fields_ = [("D", C.POINTER(C.c_double))]


class LseParams(C.Structure):
    _fields_ = [("lambda_", C.c_double)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("cgo_oracle.c", "cgo_oracle.h", "Makefile")]
    if force or not os.path.exists(_SO) or not os.path.exists(_SO_OMP) or not os.path.exists(_SO_EXACT) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


_lib = None
_use_omp = False
_use_exact = False


def use_openmp(on: bool = True) -> None:
    """Switch this process to the -fopenmp build (bench.py's all-cores baseline, the BASELINE-size parity children)."""
    global _lib, _use_omp, _use_exact
    _use_omp, _use_exact, _lib = on, False, None


def use_exact(on: bool = True) -> None:
    """Switch this process to the ARBITER build (-DORC_EXACT_SUMS -fopenmp): every reduction of the path accumulated
    in twice the working precision and rounded once; element-wise and scalar arithmetic unchanged (cgo_oracle.c header)."""
    global _lib, _use_omp, _use_exact
    _use_exact, _use_omp, _lib = on, False, None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        so = _SO_EXACT if _use_exact else (_SO_OMP if _use_omp else _SO)
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        dp = C.POINTER(C.c_double)
        L.orc_status_name.restype = C.c_char_p
        L.orc_status_name.argtypes = [C.c_int]
        L.orc_minimizeobjective.restype = C.c_int
        L.orc_minimizeobjective.argtypes = [
            C.c_void_p, C.c_void_p, dp, C.c_int64, C.POINTER(CGConfig), C.POINTER(LSConfig),
            C.POINTER(Results)]
        L.orc_minimizeobjectivererun.restype = C.c_int
        L.orc_minimizeobjectivererun.argtypes = [
            C.c_void_p, C.c_void_p, dp, C.c_int64, C.POINTER(CGConfig), C.POINTER(LSConfig),
            C.POINTER(CGConfig), C.POINTER(LSConfig), C.c_int, C.POINTER(Results), C.POINTER(C.c_int)]
        L.orc_solvesystem.restype = C.c_int
        L.orc_solvesystem.argtypes = [
            C.c_void_p, C.c_void_p, dp, C.c_int64, C.POINTER(CGConfig), C.POINTER(LSSConfig),
            C.POINTER(Results)]
        L.orc_lss_default_max_iters.restype = C.c_int64
        L.orc_lss_default_max_iters.argtypes = [C.c_double]
        L.orc_getbeta.restype = C.c_double
        L.orc_getbeta.argtypes = [C.POINTER(BetaConfig), dp, dp, dp, C.c_int64]
        L.orc_updatedir.restype = None
        L.orc_updatedir.argtypes = [dp, dp, C.c_double, C.c_int64]
        L.orc_evalwolfeconditions.restype = None
        L.orc_evalwolfeconditions.argtypes = [
            C.POINTER(LSConfig), C.c_double, C.c_double, C.c_double, dp, C.c_int64, C.c_double,
            C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_dot.restype = C.c_double
        L.orc_dot.argtypes = [dp, dp, C.c_int64]
        L.orc_norm.restype = C.c_double
        L.orc_norm.argtypes = [dp, C.c_int64]
        L.orc_exact_sums.restype = C.c_int
        L.orc_exact_sums.argtypes = []
        if _use_exact:
            for nm in ("orc_dot_f128",):
                getattr(L, nm).restype = C.c_double
                getattr(L, nm).argtypes = [dp, dp, C.c_int64]
            for nm in ("orc_sum", "orc_sum_f128"):
                getattr(L, nm).restype = C.c_double
                getattr(L, nm).argtypes = [dp, C.c_int64]
        L.orc_uniform.restype = C.c_double
        L.orc_uniform.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_fill_uniform.restype = None
        L.orc_fill_uniform.argtypes = [dp, C.c_int64, C.c_int64, C.c_uint64, C.c_double, C.c_double]
        L.orc_check_cg_config.restype = C.c_int
        L.orc_check_cg_config.argtypes = [C.POINTER(CGConfig)]
        L.orc_check_ls_config.restype = C.c_int
        L.orc_check_ls_config.argtypes = [C.POINTER(LSConfig)]
        for name in ("booth", "quad_diag", "rosenbrock_paired", "rosenbrock_chained", "lse"):
            f = getattr(L, "orc_fdf_" + name)
            f.restype = C.c_double
            f.argtypes = [C.c_void_p, dp, dp, C.c_int64]
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def beta_config(kind: str, mu: float = 0.1, m: int = 10) -> BetaConfig:
    return BetaConfig(BETA_KINDS[kind], m, mu)


def strong_wolfe(c1, c2, growth=2.0, max_iters=1000, zoom_max_iters=100) -> LSConfig:
    """setupStrongWolfeBisection (nocedal.jl:14-30)."""
    return LSConfig(0, 0, c1, c2, growth, 0.0, 0.0, max_iters, zoom_max_iters, 0, 0.0)


def wolfe_bisection(cond: str, c1, c2, delta1=0.0, max_iters=100, max_step_size=1e12,
                    feasibility_max_iters=50) -> LSConfig:
    """WolfeBisection(condition, max_iters, max_step_size, feasibility_max_iters) (wolfe.jl:6-11)."""
    return LSConfig(1, COND_KINDS[cond], c1, c2, 2.0, delta1, max_step_size, max_iters, 0,
                    feasibility_max_iters, 0.0)


def backtracking(c1, discount_factor, max_iters=100, feasibility_max_iters=50) -> LSConfig:
    """Backtracking(Armijo(c1), discount_factor, max_iters, feasibility_max_iters) (geometric.jl:15-20,159-162)."""
    return LSConfig(2, 2, c1, 0.0, 2.0, 0.0, 0.0, max_iters, 0, feasibility_max_iters, discount_factor)


def cg_config(eps, beta: BetaConfig, max_iters=1000, trace=True) -> CGConfig:
    """setupCGConfig (types.jl:171-203)."""
    return CGConfig(eps, beta, max_iters, 0, 1 if trace else 0)


@dataclass
class Objective:
    """A built-in objective bound to its parameter block (keeps buffers alive)."""
    name: str
    fn: object
    user: object = None
    keep: list = field(default_factory=list)

    @property
    def user_ptr(self):
        return C.cast(C.pointer(self.user), C.c_void_p) if self.user is not None else None

    def __call__(self, x: np.ndarray):
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.empty_like(x)
        f = self.fn(self.user_ptr, _dp(g), _dp(x), x.size)
        return f, g


def objective(name: str, D: np.ndarray | None = None, lam: float = 0.0) -> Objective:
    L = lib()
    fn = getattr(L, "orc_fdf_" + name)
    if name == "quad_diag":
        D = np.ascontiguousarray(D, dtype=np.float64)
        return Objective(name, fn, QuadParams(_dp(D)), [D])
    if name == "lse":
        return Objective(name, fn, LseParams(lam))
    return Objective(name, fn)


def python_objective(pyfdf) -> Objective:
    """Wrap a Python fdf(g, x) -> f callable (the reference's closure contract)."""
    def tramp(_user, gp, xp, n):
        g = np.ctypeslib.as_array(gp, shape=(n,))
        x = np.ctypeslib.as_array(xp, shape=(n,))
        return float(pyfdf(g, x))
    cb = FDF_T(tramp)
    return Objective("python", cb, None, [cb, pyfdf])


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
    total_fdf_evals: int
    reference_throws: bool = False  # solvesystem only: the reference raises UndefVarError here
    log_margin: np.ndarray = None   # per logged evaluation: smallest relative margin of the branches decided on it
    snap_iters: np.ndarray = None   # checkpoints reached (outer iteration numbers) …
    snap_x: np.ndarray = None       # … and the iterate after each, [len(snap_iters), n]
    trace_time: np.ndarray = None   # CLOCK_MONOTONIC seconds at the end of every outer iteration


class _Bufs:
    def __init__(self, n, max_iters, log_cap, snap_iters=None):
        self.n = n
        self.snap_iters = np.ascontiguousarray(sorted(snap_iters), dtype=np.int64) if snap_iters is not None and len(snap_iters) else None
        self.snap_x = np.zeros((len(self.snap_iters), n)) if self.snap_iters is not None else None
        self.minimizer = np.empty(n)
        self.gradient = np.empty(n)
        t = max(int(max_iters), 1)
        self.to = np.zeros(t); self.tg = np.zeros(t); self.ts = np.zeros(t); self.tt = np.zeros(t)
        self.te = np.zeros(t, dtype=np.int64)
        lc = max(int(log_cap), 1)
        self.la = np.zeros(lc); self.lp = np.zeros(lc); self.ld = np.zeros(lc); self.lm = np.zeros(lc)
        self.log_cap = int(log_cap)

    def fill(self, r: Results):
        r.minimizer = _dp(self.minimizer); r.gradient = _dp(self.gradient)
        r.trace_objective = _dp(self.to); r.trace_grad_norm = _dp(self.tg)
        r.trace_step_size = _dp(self.ts)
        r.trace_objective_evals = self.te.ctypes.data_as(C.POINTER(C.c_int64))
        r.log_cap = self.log_cap
        r.log_a = _dp(self.la); r.log_phi = _dp(self.lp); r.log_dphi = _dp(self.ld)
        r.log_margin = _dp(self.lm)
        r.trace_time = _dp(self.tt)
        if self.snap_iters is not None:
            r.nsnap = len(self.snap_iters)
            r.snap_iters = self.snap_iters.ctypes.data_as(C.POINTER(C.c_int64))
            r.snap_x = _dp(self.snap_x)

    def out(self, r: Results) -> Out:
        k = int(r.iters_ran)
        ll = min(int(r.log_len), self.log_cap)
        return Out(r.objective, self.minimizer, self.gradient, k, STATUS_NAMES[r.status],
                   self.to[:k].copy(), self.tg[:k].copy(), self.ts[:k].copy(), self.te[:k].copy(),
                   self.la[:ll].copy(), self.lp[:ll].copy(), self.ld[:ll].copy(),
                   int(r.total_fdf_evals), log_margin=self.lm[:ll].copy(),
                   snap_iters=self.snap_iters[:int(r.snap_done)].copy() if self.snap_iters is not None else None,
                   snap_x=self.snap_x[:int(r.snap_done)] if self.snap_iters is not None else None,
                   trace_time=self.tt[:k].copy())


def minimizeobjective(obj: Objective, x0, cfg: CGConfig, ls: LSConfig, log_cap: int = 0, snap_iters=None) -> Out:
    L = lib()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    b = _Bufs(x0.size, cfg.max_iters, log_cap, snap_iters)
    r = Results()
    b.fill(r)
    fnp = C.cast(obj.fn, C.c_void_p)
    rc = L.orc_minimizeobjective(fnp, obj.user_ptr, _dp(x0), x0.size, C.byref(cfg), C.byref(ls),
                                 C.byref(r))
    if rc != 0:
        raise AssertionError(f"oracle config assertion failed (code {rc})")
    return b.out(r)


def minimizeobjectivererun(obj: Objective, x0, cfg, ls, *pairs, log_cap: int = 0):
    L = lib()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    npairs = len(pairs)
    rets = (Results * (1 + npairs))()
    allcfg = [cfg] + [p[0] for p in pairs]
    bufs = []
    for i in range(1 + npairs):
        b = _Bufs(x0.size, allcfg[i].max_iters, log_cap)
        b.fill(rets[i])
        bufs.append(b)
    cfgs = (CGConfig * max(npairs, 1))(*[p[0] for p in pairs])
    lss = (LSConfig * max(npairs, 1))(*[p[1] for p in pairs])
    nrets = C.c_int(0)
    fnp = C.cast(obj.fn, C.c_void_p)
    rc = L.orc_minimizeobjectivererun(fnp, obj.user_ptr, _dp(x0), x0.size, C.byref(cfg),
                                      C.byref(ls), cfgs, lss, npairs, rets, C.byref(nrets))
    if rc != 0:
        raise AssertionError(f"oracle config assertion failed (code {rc})")
    return [bufs[i].out(rets[i]) for i in range(nrets.value)]


def linesearch_solve_sys(s, sigma=0.5, rho=0.95, max_iters=None) -> LSSConfig:
    """setupLinesearchSolveSys (solve_system.jl:13-27)."""
    if max_iters is None:
        max_iters = lib().orc_lss_default_max_iters(rho)
    return LSSConfig(rho, sigma, s, int(max_iters))


def solvesystem(obj: Objective, x0, cfg: CGConfig, ls: LSSConfig, log_cap: int = 0) -> Out:
    """solve_system.jl:64-253.  Out.reference_throws: the reference raises UndefVarError at this point."""
    L = lib()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    b = _Bufs(x0.size, cfg.max_iters, log_cap)
    r = Results()
    b.fill(r)
    fnp = C.cast(obj.fn, C.c_void_p)
    rc = L.orc_solvesystem(fnp, obj.user_ptr, _dp(x0), x0.size, C.byref(cfg), C.byref(ls), C.byref(r))
    if rc != 0:
        raise AssertionError(f"oracle config assertion failed (code {rc})")
    out = b.out(r)
    out.reference_throws = bool(r._pad)
    return out


def getbeta(kind: str, g_next, g, u, mu=0.1) -> float:
    L = lib()
    gn = np.ascontiguousarray(g_next, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    b = beta_config(kind, mu)
    return L.orc_getbeta(C.byref(b), _dp(gn), _dp(g), _dp(u), gn.size)


def evalwolfeconditions(ls: LSConfig, phi_a, dphi_a, a, u, phi_0, dphi_0):
    L = lib()
    u = np.ascontiguousarray(u, dtype=np.float64)
    v1, v2 = C.c_int(0), C.c_int(0)
    L.orc_evalwolfeconditions(C.byref(ls), phi_a, dphi_a, a, _dp(u), u.size, phi_0, dphi_0,
                              C.byref(v1), C.byref(v2))
    return bool(v1.value), bool(v2.value)


def fill_uniform(n: int, seed: int, lo: float, hi: float, offset: int = 0) -> np.ndarray:
    v = np.empty(n)
    lib().orc_fill_uniform(_dp(v), offset, n, seed, lo, hi)
    return v
