"""numpy restatement of the reference hot path — SECOND, independently written oracle.

TEST INFRASTRUCTURE ONLY (see oracle/cgo_oracle.h).  Written from the reference's
Julia text, not from cgo_oracle.c, so that a transcription slip in either shows up
as a disagreement (tests/test_oracle.py compares the two; tests/golden/ is
generated from this file by tests/golden/make_golden.py and must be reproduced
by the C oracle).  Uses np.dot / scipy dnrm2, i.e. the same OpenBLAS family that
Julia's LinearAlgebra.dot / norm dispatch to.

Citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

try:  # BLAS.nrm2 is what LinearAlgebra.norm calls for length ≥ 32
    from scipy.linalg.blas import dnrm2 as _dnrm2
except Exception:  # pragma: no cover
    _dnrm2 = None


def dot(a, b):
    return float(np.dot(a, b))


def norm(a):
    """LinearAlgebra.norm: BLAS.nrm2 for length ≥ 32, generic_norm2 (scaled when needed) below."""
    if _dnrm2 is not None and a.size >= 32:
        return float(_dnrm2(a))
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        ss = float(np.dot(a, a))
        if 1e-280 <= ss <= 1e300:
            return math.sqrt(ss)
        if np.any(np.isnan(a)):
            return math.nan
        m = float(np.max(np.abs(a))) if a.size else 0.0
        if m == 0.0 or math.isinf(m):
            return m
        r = a / m
        return m * math.sqrt(float(np.dot(r, r)))


def fdiv(a, b):
    """IEEE-754 division like Julia's `/` (±Inf / NaN instead of Python's ZeroDivisionError)."""
    with np.errstate(divide="ignore", invalid="ignore", over="ignore", under="ignore"):
        return float(np.float64(a) / np.float64(b))


def jl_max(*v):
    """Base.max — NaN-propagating (cg_flavours.jl:68)."""
    return math.nan if any(math.isnan(t) for t in v) else max(v)


def jl_min(*v):
    return math.nan if any(math.isnan(t) for t in v) else min(v)


# ---------------------------------------------------------------- configs
@dataclass
class HagerZhang:  # cg_flavours.jl:83
    pass


@dataclass
class YuanWangSheng:  # cg_flavours.jl:46-48
    mu: float = 0.1


@dataclass
class SallehAlhawarat:  # cg_flavours.jl:130
    pass


@dataclass
class LiuStorrey:  # cg_flavours.jl:154
    pass


@dataclass
class PolakRibiere:  # NEW
    pass


@dataclass
class HestenesStiefel:  # NEW (commented at cg_flavours.jl:110-127)
    pass


@dataclass
class DaiYuan:  # NEW
    pass


@dataclass
class LBFGS:  # NEW QNβConfig
    m: int = 10
    S: list = field(default_factory=list)
    Y: list = field(default_factory=list)
    rho: list = field(default_factory=list)
    gamma: float = 1.0


@dataclass
class StrongWolfeBisection:  # nocedal.jl:3-11
    c1: float
    c2: float
    a_max_growth_factor: float = 2.0
    max_iters: int = 1000
    zoom_max_iters: int = 100


@dataclass
class Wolfe:  # wolfe.jl:259-262
    c1: float
    c2: float


@dataclass
class YuanWeiLuWolfe:  # wolfe.jl:213-217
    c1: float
    c2: float
    delta1: float


@dataclass
class WolfeBisection:  # wolfe.jl:6-11
    condition: object
    max_iters: int
    max_step_size: float
    feasibility_max_iters: int


@dataclass
class Armijo:  # geometric.jl:159-162
    c1: float


@dataclass
class Backtracking:  # geometric.jl:15-20
    condition: object
    discount_factor: float
    max_iters: int
    feasibility_max_iters: int


@dataclass
class CGConfig:  # types.jl:156-169
    eps: float
    beta_config: object
    max_iters: int = 1000
    trace: bool = True


@dataclass
class Results:  # types.jl:107-114 (+ trace arrays of types.jl:17-23)
    objective: float
    minimizer: np.ndarray
    gradient: np.ndarray
    iters_ran: int
    status: str
    trace_objective: list
    trace_grad_norm: list
    trace_step_size: list
    trace_objective_evals: list
    log: list  # (a, ϕ, dϕ) of every evalϕdϕ! call


class _Info:  # LineSearchContainer, types.jl:84-100
    def __init__(self, n):
        self.xp = np.empty(n)
        self.df_xp = np.empty(n)
        self.x = np.empty(n)
        self.u = np.empty(n)
        self.log = []


# ---------------------------------------------------------------- cg_utils.jl:4-23
def evalphidphi(info, fdf, a):
    info.xp[:] = info.x + a * info.u
    phi = fdf(info.df_xp, info.xp)
    dphi = dot(info.df_xp, info.u)
    info.log.append((a, phi, dphi))
    return phi, dphi


# ---------------------------------------------------------------- getβ
def getbeta(cfg, gn, g, u):
    if isinstance(cfg, YuanWangSheng):  # cg_flavours.jl:51-79
        y = gn - g
        R1 = cfg.mu * norm(u) * norm(y)
        R2 = dot(u, y)
        R3 = fdiv(2 * dot(y, y) * dot(u, gn), dot(y, gn))
        R = jl_max(R1, R2, R3)
        with np.errstate(all="ignore"):
            tmp2 = gn / np.float64(R)
            m = fdiv(2 * dot(y, y), R)
            tmp1 = y - m * u
            return dot(tmp1, tmp2)
    if isinstance(cfg, HagerZhang):  # cg_flavours.jl:87-108
        y = gn - g
        R = dot(u, y)
        with np.errstate(all="ignore"):
            tmp2 = gn / np.float64(R)
            m = fdiv(2 * dot(y, y), R)
            tmp1 = y - m * u
            return dot(tmp1, tmp2)
    if isinstance(cfg, SallehAlhawarat):  # cg_flavours.jl:133-151
        norm_sq = norm(gn) ** 2
        tmp = dot(gn, g)
        if norm_sq > tmp:
            return fdiv(norm_sq - tmp, dot(u, gn) - dot(u, g))
        return 0.0
    if isinstance(cfg, LiuStorrey):  # cg_flavours.jl:157-170
        y = gn - g
        return fdiv(dot(gn, y), -dot(u, y))
    if isinstance(cfg, HestenesStiefel):
        y = gn - g
        return fdiv(dot(gn, y), dot(u, y))
    if isinstance(cfg, PolakRibiere):
        y = gn - g
        return fdiv(dot(gn, y), dot(g, g))
    if isinstance(cfg, DaiYuan):
        y = gn - g
        return fdiv(dot(gn, gn), dot(u, y))
    raise TypeError(cfg)


def lbfgs_push(q: LBFGS, gn, g, u, a_star):
    s = a_star * u
    y = gn - g
    sy = dot(s, y)
    if not (sy > 0.0):
        return
    q.S.append(s); q.Y.append(y); q.rho.append(1.0 / sy)
    q.gamma = fdiv(sy, dot(y, y))
    if len(q.S) > q.m:
        q.S.pop(0); q.Y.pop(0); q.rho.pop(0)


def lbfgs_dir(q: LBFGS, g):
    r = g.copy()
    k = len(q.S)
    alpha = [0.0] * k
    for i in range(k - 1, -1, -1):
        alpha[i] = q.rho[i] * dot(q.S[i], r)
        r = r - alpha[i] * q.Y[i]
    r = (q.gamma if k > 0 else 1.0) * r
    for i in range(k):
        b = q.rho[i] * dot(q.Y[i], r)
        r = r + (alpha[i] - b) * q.S[i]
    return -r


# ---------------------------------------------------------------- qn_flavours.jl (dense BroydenFamily, as written)
@dataclass
class BroydenFamily:  # qn_flavours.jl:53-66; setupBroydenFamily fills B with NaN = "use the default"
    theta: float
    B: np.ndarray | None = None


def isposdef(B):  # LinearAlgebra.isposdef: Hermitian and Cholesky succeeds
    if not np.all(np.isfinite(B)) or not np.array_equal(B, B.T):
        return False
    try:
        np.linalg.cholesky(B)
        return True
    except np.linalg.LinAlgError:
        return False


def broyden_init_dir(q: BroydenFamily, g):  # initializeLineSearchContainer! (QN), qn_flavours.jl:25-44
    n = len(g)
    if q.B is None:
        q.B = np.full((n, n), np.nan)
    if not isposdef(q.B):
        q.B[:] = np.eye(n)
    return np.linalg.solve(q.B, -g)


def broyden_getbeta(q: BroydenFamily, gn, g, u):  # qn_flavours.jl:70-90
    B = q.B
    y = gn - g
    s = np.linalg.solve(B, y)                     # :81  — hence Bs = y up to rounding
    Bs = B @ s
    sBs = dot(s, Bs)
    tmp = q.theta * sBs
    with np.errstate(divide="ignore", invalid="ignore"):
        v = y / dot(s, y) - Bs / sBs              # ≈ 0
        B[:] = B - np.outer(Bs, Bs) / sBs + np.outer(y, y) / dot(s, y) + tmp * np.outer(v, v)   # ≈ B
    return B


def broyden_dir(q: BroydenFamily, g):  # updatedir!(u, df_x, B), qn_flavours.jl:3-22
    if not isposdef(q.B):
        q.B[:] = np.eye(len(g))
    return np.linalg.solve(q.B, -g)


# ---------------------------------------------------------------- nocedal.jl
def zoom(info, fdf, a_lb, a_ub, phi_lb, phi0, dphi0, c1, c2, evals, max_iters):  # :162-209
    a = phi_a = dphi_a = 0.0
    for _ in range(max_iters):
        a = (a_lb + a_ub) / 2
        phi_a, dphi_a = evalphidphi(info, fdf, a)
        evals += 1
        if (phi_a > phi0 + c1 * a * dphi0) or (phi_a >= phi_lb):
            a_ub = a
        else:
            if abs(dphi_a) <= -c2 * dphi0:
                return phi_a, a, evals, "success"
            if dphi_a * (a_ub - a_lb) >= 0:
                a_ub = a_lb
            a_lb = a
            phi_lb = phi_a
    return phi_a, a, evals, "zoom_max_iters_reached"


def linesearch_strong(info, cfg: StrongWolfeBisection, fdf, f_x, df_x, a_initial):  # :33-158
    c1, c2, growth = cfg.c1, cfg.c2, cfg.a_max_growth_factor
    if not (0.0 < a_initial and math.isfinite(a_initial)):
        a_initial = 1.0
    phi0 = f_x
    dphi0 = dot(df_x, info.u)
    if dphi0 > 0.0:
        return phi0, 0.0, 0, "non_descent_search_direction"
    a_prev, phi_prev = 0.0, phi0
    a, phi_a, dphi_a = a_initial, phi0, dphi0
    evals = 0
    non_initial = False
    for _ in range(cfg.max_iters):
        phi_a, dphi_a = evalphidphi(info, fdf, a)
        evals += 1
        chk1 = phi_a > phi0 + c1 * a * dphi0
        chk2 = phi_a >= phi_prev
        if chk1 or (chk2 and non_initial):
            return zoom(info, fdf, a_prev, a, phi_prev, phi0, dphi0, c1, c2, evals, cfg.zoom_max_iters)
        if abs(dphi_a) <= -c2 * dphi0:
            return phi_a, a, evals, "success"
        if dphi_a >= 0:
            return zoom(info, fdf, a, a_prev, phi_a, phi0, dphi0, c1, c2, evals, cfg.zoom_max_iters)
        a_prev, phi_prev, non_initial = a, phi_a, True
        a_max = a * growth
        if a > a_max:
            return phi_a, a, evals, "linesearch_a_max_overflow"
        a = (a_max + a) / 2
    return phi_a, a, evals, "linesearch_max_iters_reached"


# ---------------------------------------------------------------- wolfe.jl
def evalwolfeconditions(cond, phi_a, dphi_a, a, u, phi0, dphi0):
    if isinstance(cond, YuanWeiLuWolfe):  # :219-251
        assert 0.0 < cond.delta1 < cond.c1 < cond.c2 < 1.0
        nu = dot(u, u)
        rhs1 = phi0 + cond.c1 * a * dphi0 + a * jl_min(-cond.delta1 * dphi0, cond.c1 * a * nu / 2)
        rhs2 = cond.c2 * dphi0 + jl_min(-cond.delta1 * dphi0, cond.c1 * a * nu)
        return phi_a <= rhs1, dphi_a >= rhs2
    assert 0.0 < cond.c1 < cond.c2 < 1.0  # :278
    return phi_a <= phi0 + cond.c1 * a * dphi0, dphi_a >= cond.c2 * dphi0  # :286-291


def findfeasiblestepsize(info, fdf, evals, a, reduction, lb, max_iters):  # :171-207
    if lb > a:
        return 0.0, 0.0, a, evals, "bisection_lower_bound_larger_than_proposed_step"
    phi_a, dphi_a = evalphidphi(info, fdf, a)
    evals += 1
    it = 1
    while a > lb and it < max_iters:
        if math.isfinite(phi_a) and math.isfinite(dphi_a):
            return phi_a, dphi_a, a, evals, "feasible"
        a = a * reduction
        phi_a, dphi_a = evalphidphi(info, fdf, a)
        evals += 1
        it += 1
    return phi_a, dphi_a, a, evals, "infeasible"


def linesearch_wolfe(info, cfg: WolfeBisection, fdf, f_x, df_x, a_initial):  # :13-165
    reduction, growth = 0.5, 2.0
    u = info.u
    if not (cfg.max_step_size > a_initial > 0.0):
        a_initial = jl_min(1.0, cfg.max_step_size / 2)
    phi0 = f_x
    if not math.isfinite(phi0):
        return phi0, 0.0, 0, "accepted_non_finite_iterate"
    dphi0 = dot(df_x, u)
    if dphi0 > 0.0:
        return phi0, 0.0, 0, "non_descent_search_direction"
    a, evals, lb, ub = a_initial, 0, 0.0, math.inf
    phi_a, dphi_a, a, evals, flag = findfeasiblestepsize(
        info, fdf, evals, a, reduction, 0.0, cfg.feasibility_max_iters)
    if flag != "feasible":
        return phi0, 0.0, 0, "cannot_find_initial_feasible_step"
    for _ in range(cfg.max_iters):
        ok_large, ok_small = evalwolfeconditions(cfg.condition, phi_a, dphi_a, a, u, phi0, dphi0)
        if (not ok_large) or (not ok_small):
            if not ok_large:
                ub = a
                a = (lb + ub) / 2
            else:
                lb = a
                if not math.isfinite(ub):
                    a = growth * a
                    if a > cfg.max_step_size:
                        return phi0, 0.0, 0, "max_step_length_reached"
                else:
                    a = (lb + ub) / 2
            if not (lb < a < ub):
                if norm(u + df_x) != 0.0:  # !isapprox(norm(u+df_x), 0)  (:123)
                    lb, ub, a = 0.0, math.inf, a_initial
                    u[:] = -df_x
                # else: wolfe.jl:131 is a bare expression — no return
            phi_a, dphi_a, a, evals, flag = findfeasiblestepsize(
                info, fdf, evals, a, reduction, lb, cfg.feasibility_max_iters)
            if flag != "feasible":
                return phi0, 0.0, 0, "cannot_find_feasible_step"
        else:
            return phi_a, a, evals, "success"
    return phi_a, a, evals, "linesearch_max_iters_reached"


# ---------------------------------------------------------------- geometric.jl (bug for bug)
def evalbacktrackcondition(cond: Armijo, phi_a, a, phi0, dphi0):  # :164-186
    assert 0.0 < cond.c1 < 1.0
    if not math.isfinite(phi0) or not math.isfinite(phi_a) or not math.isfinite(a):
        return False
    return (phi0 - phi_a) >= -cond.c1 * a * dphi0


def geometricsearch(info, fdf, a, cond, max_iters, rho, divide, evals, phi_a, phi0, dphi0):  # :102-152
    a_prev, phi_prev = a, phi_a
    for _ in range(max_iters):
        a = a / rho if divide else a * rho
        if not math.isfinite(a):
            return phi_prev, a_prev, evals, "non_finite_step_proposed"
        if a == a_prev:
            return phi_prev, a_prev, evals, "proposed_step_same_as_current_step"
        phi_a, _ = evalphidphi(info, fdf, a)
        evals += 1
        if not evalbacktrackcondition(cond, phi_a, a, phi0, dphi0):
            return phi_prev, a_prev, evals, "success"  # xp/df_xp hold the REJECTED trial (:141-144)
        a_prev, phi_prev = a, phi_a
    return phi_a, a, evals, "linesearch_max_iters_reached"


def linesearch_backtracking(info, cfg: Backtracking, fdf, f_x, df_x, a_initial):  # :22-100
    phi0 = f_x
    if not math.isfinite(phi0):
        return phi0, 0.0, 0, "accepted_non_finite_iterate"
    dphi0 = dot(df_x, info.u)
    if dphi0 > 0.0:
        return phi0, 0.0, 0, "non_descent_search_direction"
    evals = 0
    a = a_initial
    if not math.isfinite(a):
        uu = dot(info.u, info.u)
        a = fdiv(abs(phi0), uu)
    if not math.isfinite(a):
        a = 1.0
    phi_a, dphi_a, a, evals, flag = findfeasiblestepsize(info, fdf, evals, a, 0.5, 0.0, cfg.feasibility_max_iters)
    if flag != "feasible":
        return phi0, 0.0, 0, "cannot_find_initial_feasible_step"
    phi_a, _ = evalphidphi(info, fdf, a)  # :77-78, redundant
    evals += 1
    valid = evalbacktrackcondition(cfg.condition, phi_a, a, phi0, dphi0)
    return geometricsearch(info, fdf, a, cfg.condition, cfg.max_iters, cfg.discount_factor, valid, evals,
                           phi_a, phi0, dphi0)


# ---------------------------------------------------------------- optim.jl:6-171
def minimizeobjective(fdf, x_initial, config: CGConfig, ls_config) -> Results:
    assert 0.0 < config.eps < 1.0  # types.jl:187
    if isinstance(ls_config, StrongWolfeBisection):
        assert 0.0 < ls_config.c1 < ls_config.c2 < 1.0 and ls_config.a_max_growth_factor > 1
    n = len(x_initial)
    bc = config.beta_config
    qn = isinstance(bc, LBFGS)
    bf = isinstance(bc, BroydenFamily)
    if qn:
        bc = LBFGS(bc.m)
    if bf:
        assert 0.0 <= bc.theta  # qn_flavours.jl:57
        bc = BroydenFamily(bc.theta)
    df_x = np.empty(n)
    x = np.array(x_initial, dtype=np.float64)
    f_x = fdf(df_x, x)
    norm_df_x = norm(df_x)
    f_x0 = f_x
    tr = ([], [], [], [])
    info = _Info(n)
    info.u[:] = broyden_init_dir(bc, df_x) if bf else -df_x
    info.x[:] = x
    info.xp[:] = x
    info.df_xp[:] = df_x
    a_initial = math.nan

    def done(i, status):
        return Results(f_x, x.copy(), df_x.copy(), i, status, tr[0][:i], tr[1][:i], tr[2][:i],
                       tr[3][:i], info.log)

    for it in range(1, config.max_iters + 1):
        if math.isfinite(f_x) and math.isfinite(norm_df_x) and norm_df_x < config.eps:
            return done(it - 1, "success" if f_x <= f_x0 else "increasing_objective")
        if isinstance(ls_config, StrongWolfeBisection):
            f_xp, a_star, evals, status = linesearch_strong(info, ls_config, fdf, f_x, df_x, a_initial)
        elif isinstance(ls_config, Backtracking):
            f_xp, a_star, evals, status = linesearch_backtracking(info, ls_config, fdf, f_x, df_x, a_initial)
        else:
            f_xp, a_star, evals, status = linesearch_wolfe(info, ls_config, fdf, f_x, df_x, a_initial)
        a_initial = a_star
        if status != "success":
            return done(it - 1, status)
        norm_df_xp = norm(info.df_xp)
        if not math.isfinite(f_xp) or not math.isfinite(norm_df_xp):
            return done(it - 1, "non_finite_objective_or_gradient_proposed")
        if qn:
            lbfgs_push(bc, info.df_xp, df_x, info.u, a_star)
        elif bf:
            broyden_getbeta(bc, info.df_xp, df_x, info.u)
        else:
            beta = getbeta(bc, info.df_xp, df_x, info.u)
        x[:] = info.xp
        f_x = f_xp
        df_x[:] = info.df_xp
        info.x[:] = x
        norm_df_x = norm_df_xp
        if qn:
            info.u[:] = lbfgs_dir(bc, df_x)
        elif bf:
            info.u[:] = broyden_dir(bc, df_x)
        else:
            info.u[:] = -df_x + beta * info.u  # cg_flavours.jl:10-12
        tr[0].append(f_x); tr[1].append(norm_df_x); tr[2].append(a_star); tr[3].append(evals)
    return done(config.max_iters, "max_iters_reached")


def minimizeobjectivererun(fdf, x_initial, config, ls_config, *pairs):  # optim.jl:173-208
    rets = [minimizeobjective(fdf, x_initial, config, ls_config)]
    for cfg_k, ls_k in pairs:
        if rets[-1].status != "success":
            rets.append(minimizeobjective(fdf, rets[-1].minimizer, cfg_k, ls_k))
        else:
            return rets
    return rets


# ---------------------------------------------------------------- solve_system.jl
@dataclass
class LinesearchSolveSys:  # solve_system.jl:6-11; defaults of setupLinesearchSolveSys :13-27
    s: float
    sigma: float = 0.5
    rho: float = 0.95
    max_iters: int | None = None

    def __post_init__(self):
        assert 0.0 < self.rho < 1.0 and self.s > 0.0  # :21-23 (σ is never checked)
        if self.max_iters is None:
            self.max_iters = int(round(math.log(1e-6) / math.log(self.rho)))  # round(Int, log(ρ, 1e-6))


def solvesystem(fdf, x_initial, config: CGConfig, ls: LinesearchSolveSys) -> Results:  # :64-237
    """Written independently of cgo_oracle.c from solve_system.jl.  Keeps the two array OBJECTS
    of the reference (`x`, `x_next`) and swaps the names, so the projection step lands on the
    iterate of two iterations ago exactly as there.  When no trial passes, the reference reads the
    loop variable outside its scope and throws; this restatement returns :linesearch_failed."""
    assert 0.0 < config.eps < 1.0
    n = len(x_initial)
    bc = config.beta_config
    df_x = np.empty(n)
    x = np.array(x_initial, dtype=np.float64)
    x_next = np.array(x_initial, dtype=np.float64)
    f_x = fdf(df_x, x)
    norm_df_x = norm(df_x)
    tr = ([], [], [], [])
    info = _Info(n)
    info.u[:] = -df_x
    info.x[:] = x
    info.xp[:] = x
    info.df_xp[:] = df_x

    def done(xx, gg, f, i, status):
        return Results(f, xx.copy(), gg.copy(), i, status, tr[0][:i], tr[1][:i], tr[2][:i], tr[3][:i], info.log)

    for it in range(1, config.max_iters + 1):
        if norm_df_x < config.eps:
            return done(x, df_x, f_x, it - 1, "success")
        norm_u_sq = dot(info.u, info.u)
        hit = None
        for i in range(ls.max_iters):
            a = ls.s * ls.rho ** i
            f_xp, dphi = evalphidphi(info, fdf, a)
            norm_df_xp = norm(info.df_xp)
            if not (-dphi < ls.sigma * a * norm_df_xp * norm_u_sq):
                hit = i
                break
        if hit is None:
            return done(x, df_x, f_x, it - 1, "linesearch_failed")
        a_star = a
        if norm_df_xp < config.eps:
            tr[0].append(f_xp); tr[1].append(norm(info.df_xp)); tr[2].append(a_star); tr[3].append(hit)
            return done(info.xp, info.df_xp, f_xp, it, "success")
        m = fdiv(a_star * dot(info.df_xp, info.u), norm_df_xp * norm_df_xp)  # x^2 lowers to x*x (literal_pow)
        x_next[:] = x_next + m * info.df_xp
        f_x_next = fdf(info.df_xp, x_next)
        if not math.isfinite(f_x_next) or not math.isfinite(norm(info.df_xp)):
            return done(x, df_x, f_x, it - 1, "non_finite_objective_or_gradient_proposed")
        x, x_next = x_next, x
        f_x = f_x_next
        beta = getbeta(bc, info.df_xp, df_x, info.u)
        df_x[:] = info.df_xp
        info.x[:] = x
        norm_df_x = norm(df_x)
        info.u[:] = -df_x + beta * info.u
        tr[0].append(f_x); tr[1].append(norm_df_x); tr[2].append(a_star); tr[3].append(hit)
    return done(x, df_x, f_x, config.max_iters, "max_iters_reached")


# ---------------------------------------------------------------- primal_barrier.jl
@dataclass
class PrimalBarrierConfig:  # primal_barrier.jl:130-136; setupPrimalBarrierConfig :138-154 (inf_f0_lb = 0)
    barrier_tol: float
    barrier_growth_factor: float
    max_iters: int
    t_initial: float = math.nan
    inf_f0_lb: float = 0.0


@dataclass
class PrimalBarrierResults:  # primal_barrier.jl:1-7
    centering_results: list
    status: str
    iters_ran: int
    t_final: float
    total_objective_evals: int


class CvxInequalityConstraint:  # primal_barrier.jl:38-60
    def __init__(self, M, D):
        self.fi_evals = np.empty(M)
        self.dfi_evals = [np.empty(D) for _ in range(M)]
        self.grad = np.empty(D)


def evalconstraints(con, hdh, x):  # primal_barrier.jl:70-94
    hdh(con.fi_evals, con.dfi_evals, x)
    fi = con.fi_evals
    fi[fi > 0.0] = 0.0                       # clamp!(fi_evals, -Inf, 0)  (NaN stays NaN)
    with np.errstate(divide="ignore", invalid="ignore"):
        psi = -float(sum(math.log(-v) if -v > 0.0 else (-math.inf if v == 0.0 else math.nan) for v in fi))  # :82
        con.grad[:] = 0.0
        for i in range(len(con.dfi_evals)):  # :85-89 — every constraint touches every coordinate
            con.grad -= con.dfi_evals[i] / fi[i]
    return psi


def evalbarrier(con, df_x, fdf, hdh, x, t):  # primal_barrier.jl:111-128
    f_x = fdf(df_x, x)
    psi = evalconstraints(con, hdh, x)
    df_x[:] = t * df_x + con.grad
    return t * f_x + psi


def verifyt0(t0, x0, f0df0, mu, inf_f0_lb):  # primal_barrier.jl:259-276
    if not math.isfinite(t0) or t0 < 0.0:
        g = np.empty(len(x0))
        return (f0df0(g, np.array(x0, dtype=np.float64)) - inf_f0_lb) * mu
    return t0


def primalbarriermethod(con, f0df0, hdh, x_initial, centering_config, ls_config, barrier_config, *reruns):
    """primal_barrier.jl:156-255 (algorithm 11.1 of Boyd 2004) as written: `x` is copied from x_initial
    and never updated, so EVERY centering step restarts from x_initial (:172, :214-220)."""
    bc = barrier_config
    x = np.array(x_initial, dtype=np.float64)
    rets = []

    def assemble(status, it, t):  # :9-35
        total = sum(int(e) for rr in rets[:it] for r in rr for e in r.trace_objective_evals)
        return PrimalBarrierResults(rets[:it], status, it, t, total)

    hdh(con.fi_evals, con.dfi_evals, x)
    if np.any(con.fi_evals >= 0.0):          # :181-191
        return assemble("infeasible_start", 0, bc.t_initial)
    t = verifyt0(bc.t_initial, x_initial, f0df0, bc.barrier_growth_factor, bc.inf_f0_lb)  # :193
    tbox = [t]
    fdf = lambda gg, xx: evalbarrier(con, gg, f0df0, hdh, xx, tbox[0])  # noqa: E731  (:199-206)
    nc = len(con.fi_evals)
    for i in range(1, bc.max_iters + 1):     # :208
        rets.append(minimizeobjectivererun(fdf, x, centering_config, ls_config, *reruns))
        if rets[-1][-1].status != "success":
            return assemble("centering_step_issue", i, tbox[0])
        if nc / tbox[0] < bc.barrier_tol:    # :229
            return assemble("success", i, tbox[0])
        tbox[0] = bc.barrier_growth_factor * tbox[0]
    return assemble("max_iters_reached", bc.max_iters, tbox[0])


def make_boxhdh(lbs, ubs):  # examples/constrained.jl:18-48: rows 1..D upper bounds, D+1..2D lower bounds
    lbs, ubs = np.asarray(lbs, dtype=np.float64), np.asarray(ubs, dtype=np.float64)

    def hdh(fi, dfi, x):
        D = len(x)
        fi[:D] = x - ubs
        fi[D:] = lbs - x
        for d in range(D):
            dfi[d][:] = 0.0
            dfi[d][d] = 1.0
            dfi[d + D][:] = 0.0
            dfi[d + D][d] = -1.0
    return hdh


# ---------------------------------------------------------------- objectives (fdf!(g, x) -> f)
def booth(g, p):  # test_funcs.jl:3-12
    x, y = p
    g[0] = 2 * (x + 2 * y - 7) + 2 * (2 * x + y - 5) * 2
    g[1] = 2 * (x + 2 * y - 7) * 2 + 2 * (2 * x + y - 5)
    return (x + 2 * y - 7) ** 2 + (2 * x + y - 5) ** 2


def make_quad_diag(D):
    def fdf(g, x):
        g[:] = D * x
        return float(np.sum(0.5 * (g * x)))
    return fdf


def rosenbrock_paired(g, x):
    a, b = x[0::2], x[1::2]
    t1 = b - a * a
    t2 = 1.0 - a
    g[0::2] = -400.0 * (a * t1) - 2.0 * t2
    g[1::2] = 200.0 * t1
    return float(np.sum(100.0 * (t1 * t1) + t2 * t2))


def rosenbrock_chained(g, x):  # value: test_funcs.jl:50-57
    g[:] = 0.0
    t2 = 1.0 - x[:-1]
    t1 = x[1:] - x[:-1] ** 2
    g[:-1] += -2.0 * t2 - 400.0 * (x[:-1] * t1)
    g[1:] += 200.0 * t1
    return float(np.sum(t2 * t2 + 100.0 * (t1 * t1)))


def make_lse(lam):
    def fdf(g, x):
        m = float(np.max(x))
        e = np.exp(x - m)
        s = float(np.sum(e))
        g[:] = e / s + lam * x
        return (m + math.log(s)) + 0.5 * lam * dot(x, x)
    return fdf


def uniform(seed: int, idx: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser of (seed XOR index) → [0,1); same stream as orc_uniform."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) ^ idx.astype(np.uint64)) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
