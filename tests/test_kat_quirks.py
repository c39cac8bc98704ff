"""Hand-derived known answers for the paths that only "two restatements by the same author" pinned in round 1
(VERDICT r01, weak #1): every number below is derived BY HAND from the cited reference lines, with exactly
representable values, and must be reproduced by all four implementations —

    oracle/cgo_oracle.c  ·  oracle/cgo_oracle_np.py  ·  the product's host engine over the test double (CPU tier)
    ·  the product itself, libcgo_hip.so on the GPU (GPU tier, host-closure objective → cgo_objective_create_callback).

The objectives are 2-element CLOSURES `f = fdf!(g, x)` (the reference's own contract, src/cg_utils.jl:19) with a
discontinuity placed so that the line searches take the rare branches after a handful of exactly computable steps:

  wall(T):    f = r(x1) − x2 + 1024·[x2 > T],  r(t) = ½t² for t ≥ 0 else 0;   ∇f = (max(x1, 0), −1)
  barrier:    f = −x2 for x2 ≤ 0.75, +Inf beyond;                              ∇f = (0, −1)
  quad1:      f = ½x1²;                                                        ∇f = (x1, 0)
  plateau:    f = x1² for x1 ≥ 0.875, 100 below;                               ∇f = (2x1, 0) resp. (0, 0)
  nanpit:     f = x1² for x1 ≥ 0.5, NaN below;                                 ∇f = (2x1, 0) resp. (0, 0)

All closures are pure functions of x (the engines may evaluate speculative extra points).
"""
import numpy as np
import pytest

from _cases import Case, run_hostsim, run_numpy, run_oracle

C1, C2 = 2.0 ** -10, 0.875      # Wolfe(c1, c2): exactly representable
BISECT = [1.0 + 2.0 ** -k for k in range(1, 53)]   # 1.5, 1.25, …, 1 + 2^-52


def wall(T):
    def fdf(g, x):
        g[0] = x[0] if x[0] >= 0.0 else 0.0
        g[1] = -1.0
        r = 0.5 * (x[0] * x[0]) if x[0] >= 0.0 else 0.0
        return (r - x[1]) + (1024.0 if x[1] > T else 0.0)
    return fdf


def barrier(g, x):
    g[0], g[1] = 0.0, -1.0
    return -x[1] if x[1] <= 0.75 else float("inf")


def quad1(g, x):
    g[0], g[1] = x[0], 0.0
    return 0.5 * (x[0] * x[0])


def plateau(g, x):
    if x[0] >= 0.875:
        g[0], g[1] = 2.0 * x[0], 0.0
        return x[0] * x[0]
    g[0], g[1] = 0.0, 0.0
    return 100.0


def nanpit(g, x):
    if x[0] >= 0.5:
        g[0], g[1] = 2.0 * x[0], 0.0
        return x[0] * x[0]
    g[0], g[1] = 0.0, 0.0
    return float("nan")


def closure_case(name, fdf, x0, **kw):
    return Case(name, "closure", 2, np.array(x0, dtype=np.float64), extra={"fdf": fdf}, **kw)


WOLFE = dict(ls="WolfeBisection", cond="Wolfe", c1=C1, c2=C2, ls_max_iters=200, max_step_size=1e12)

KATS = {}

# ---------------------------------------------------------------------------------------------------------
# A. WolfeBisection bracket collapse on u ≠ −g → restart from steepest descent, dϕ₀ NOT recomputed (wolfe.jl:122-130)
#  it 1 (optim.jl:25-47, wolfe.jl:30-78): x0 = (1, −1): f = 1.5, g = (1, −1), u = (−1, 1), dϕ₀ = −2.  a_initial = NaN →
#     min(1, max_step/2) = 1: xp = (0, 0), ϕ = 0, dϕ = (0, −1)·(−1, 1) = −1.  ϕ ≤ 1.5 + c1·1·(−2) ✓, dϕ ≥ c2·(−2) = −1.75 ✓
#     → :success, a* = 1, 1 evaluation.  x1 = (0, 0), f = 0, g1 = (0, −1), ‖g1‖ = 1.
#     DaiYuan: y = g1 − g0 = (−1, 0), u·y = 1, β = g1·g1 / u·y = 1;  u1 = −g1 + β·u0 = (−1, 2), dϕ₀ = g1·u1 = −2.
#  it 2: a = a_initial = 1: xp = (−1, 2): r = 0, 2 is NOT > T = 2 → ϕ = −2, dϕ = (0, −1)·(−1, 2) = −2: ϕ ≤ c1·(−2) ✓ but
#     dϕ ≥ −1.75 ✗ → step too short: lb = 1, ub = ∞ → a = 2 (wolfe.jl:96-104): x2 = 4 > 2 → ϕ = 1020 ✗ → ub = 2, a = 1.5
#     (:81-95) → x2 = 3 → wall … every a_k = 1 + 2^-k has x2 = 2 + 2^(1−k) > 2 (exact for k ≤ 52) → ub = a_k.  After
#     k = 52: a = (1 + (1 + 2^-52))/2: the sum 2 + 2^-52 ties to even = 2 → a = 1 = lb → !(lb < a < ub) (:122).
#     norm(u + g1) = ‖(−1, 1)‖ ≠ 0 → lb = 0, ub = ∞, a = a_initial = 1, u ← −g1 = (0, 1) (:125-129); dϕ₀ stays −2.
#     findfeasiblestepsize!(a = 1, lb = 0) (:141): xp = (0, 1), ϕ = −1, dϕ = −1.  Next check (:70-78) with the STALE
#     dϕ₀ = −2: ϕ ≤ 0 + c1·(−2) ✓, dϕ = −1 ≥ c2·(−2) = −1.75 ✓ → :success, a* = 1, evaluations 1 + 1 + 52 + 1 = 55.
#     (A recomputed dϕ₀ = g1·(0, 1) = −1 would give −1 ≥ −0.875 ✗ and the search would go on to a = 2, 4, ….)
#     optim.jl:136-141: x2 = x1 + 1·(0, 1) = (0, 1), f = −1, g2 = (0, −1), ‖g2‖ = 1; n = max_iters = 2 → :max_iters_reached.
KATS["A-wolfe-reset"] = dict(
    case=closure_case("kat-wolfe-reset", wall(2.0), [1.0, -1.0], beta="DaiYuan", eps=1e-9, max_iters=2, **WOLFE),
    status="max_iters_reached", iters=2, objective=-1.0, minimizer=[0.0, 1.0], gradient=[0.0, -1.0],
    trace_f=[0.0, -1.0], trace_g=[1.0, 1.0], trace_a=[1.0, 1.0], trace_e=[1, 55],
    log_a=[1.0] + [1.0, 2.0] + BISECT + [1.0])

# ---------------------------------------------------------------------------------------------------------
# B1. The same collapse while u ≡ −g: wolfe.jl:131 builds a tuple and drops it (missing `return`), so the loop goes on
#  into findfeasiblestepsize!(a = lb) whose `while a > lb` (:195) never runs → :infeasible although ϕ is finite →
#  :cannot_find_feasible_step (:141-158).  wall(T = 1), x0 = (0, 0): f = 0, g = (0, −1), u = (0, 1), dϕ₀ = −1.
#  a = 1: xp = (0, 1), 1 is NOT > 1 → ϕ = −1, dϕ = −1: ϕ ≤ c1·(−1) ✓, dϕ ≥ −0.875 ✗ → lb = 1, a = 2 → wall → ub = 2, then
#  a_k = 1 + 2^-k (x2 = a_k > 1: wall) for k = 1..52, then a = 1 = lb: collapse, u ≡ −g → fall through →
#  findfeasiblestepsize!(1, lb = 1): one evaluation (the 55th), loop skipped → :infeasible.  optim.jl:93-104: last good
#  iterate x0, iters_ran = 0.
KATS["B1-feasible-a-eq-lb"] = dict(
    case=closure_case("kat-a-eq-lb", wall(1.0), [0.0, 0.0], beta="DaiYuan", eps=1e-9, max_iters=5, **WOLFE),
    status="cannot_find_feasible_step", iters=0, objective=0.0, minimizer=[0.0, 0.0], gradient=[0.0, -1.0],
    trace_f=[], trace_g=[], trace_a=[], trace_e=[], log_a=[1.0, 2.0] + BISECT + [1.0])

# ---------------------------------------------------------------------------------------------------------
# B2. findfeasiblestepsize!'s loop test `iter < max_iters` (wolfe.jl:195) comes BEFORE the finiteness test of the value
#  just computed: with feasibility_max_iters = 2 the halved step a = 0.5 is evaluated (finite: −0.5) and still reported
#  :infeasible → :cannot_find_initial_feasible_step (:51-65).  barrier, x0 = (0, 0), u = (0, 1): a = 1 → x2 = 1 > 0.75 → +Inf.
KATS["B2a-last-halving"] = dict(
    case=closure_case("kat-last-halving", barrier, [0.0, 0.0], beta="DaiYuan", eps=1e-9, max_iters=5,
                      **dict(WOLFE, feas_max_iters=2)),
    status="cannot_find_initial_feasible_step", iters=0, objective=0.0, minimizer=[0.0, 0.0], gradient=[0.0, -1.0],
    trace_f=[], trace_g=[], trace_a=[], trace_e=[], log_a=[1.0, 0.5])
#  With feasibility_max_iters = 3 the same halving IS accepted (iter = 2 < 3 → :feasible at a = 0.5: ϕ = −0.5 ≤ c1·0.5·(−1) ✓,
#  dϕ = −1 ≥ −0.875 ✗ → lb = 0.5, a = 1).  findfeasiblestepsize!(1, lb = 0.5): +Inf → a = 0.5 → finite, but now
#  `a > lb` is 0.5 > 0.5 ✗ → :infeasible → :cannot_find_feasible_step.
KATS["B2b-halving-onto-lb"] = dict(
    case=closure_case("kat-halving-onto-lb", barrier, [0.0, 0.0], beta="DaiYuan", eps=1e-9, max_iters=5,
                      **dict(WOLFE, feas_max_iters=3)),
    status="cannot_find_feasible_step", iters=0, objective=0.0, minimizer=[0.0, 0.0], gradient=[0.0, -1.0],
    trace_f=[], trace_g=[], trace_a=[], trace_e=[], log_a=[1.0, 0.5, 1.0, 0.5])

# ---------------------------------------------------------------------------------------------------------
# D. Backtracking returns the PREVIOUS (ϕ, a) while info.xp / info.df_xp hold the last, rejected trial (geometric.jl:141-144),
#  and optim.jl:136-141 adopts that trial.  quad1, x0 = (1, 0), Armijo(c1 = 0.25), discount 0.5, DaiYuan, max_iters = 2.
#  it 1: ϕ₀ = 0.5, g = (1, 0), u = (−1, 0), dϕ₀ = −1, u·u = 1.  a_initial = NaN → a = |ϕ₀|/u·u = 0.5 (geometric.jl:49-52).
#     findfeasiblestepsize!(0.5): x = 0.5, ϕ = 0.125 (1 evaluation); the re-evaluation at :77 counts too (2).  Armijo:
#     ϕ₀ − ϕ = 0.375 ≥ −c1·a·dϕ₀ = 0.125 ✓ → grow.  a = 1: x = 0, ϕ = 0 (3): 0.5 ≥ 0.25 ✓.  a = 2: x = −1, ϕ = 0.5 (4):
#     0 ≥ 0.5 ✗ → return (ϕ = 0, a = 1, 4, :success) — but xp = x0 + 2u = (−1, 0), df_xp = (−1, 0).
#     optim.jl: f = 0 (!), x1 = (−1, 0), g1 = (−1, 0), ‖g1‖ = 1; trace step 1.
#     DaiYuan on (g⁺ = (−1, 0), g = (1, 0), u = (−1, 0)): y = (−2, 0), u·y = 2, β = 1/2; u1 = (1, 0) + ½(−1, 0) = (0.5, 0), dϕ₀ = −0.5.
#  it 2: ϕ₀ = f = 0 (true f(x1) is 0.5), a = a_initial = 1: x = −0.5, ϕ = 0.125 (1, re-evaluation 2).  Armijo: 0 − 0.125 ≥
#     0.25·1·0.5 ✗ → shrink.  a = 0.5: x = −0.75, ϕ = 0.28125 (3): ✗ → return (ϕ = 0.125, a = 1, 3, :success) with
#     xp = (−0.75, 0): the shrink branch returns an Armijo-violating step.  f = 0.125, x2 = (−0.75, 0), ‖g2‖ = 0.75.
KATS["D-backtracking-adopts-rejected-trial"] = dict(
    case=closure_case("kat-backtracking", quad1, [1.0, 0.0], beta="DaiYuan", eps=1e-9, max_iters=2,
                      ls="Backtracking", c1=0.25, discount=0.5, ls_max_iters=100, feas_max_iters=50),
    status="max_iters_reached", iters=2, objective=0.125, minimizer=[-0.75, 0.0], gradient=[-0.75, 0.0],
    trace_f=[0.0, 0.125], trace_g=[1.0, 0.75], trace_a=[1.0, 1.0], trace_e=[4, 3],
    log_a=[0.5, 0.5, 1.0, 2.0, 1.0, 1.0, 0.5])

# ---------------------------------------------------------------------------------------------------------
# E. :increasing_objective (optim.jl:67-78) through the same Backtracking branch: plateau, x0 = (1, 0): ϕ₀ = 1, g = (2, 0),
#  u = (−2, 0), dϕ₀ = −4, u·u = 4 → a = 0.25: x = 0.5 < 0.875 → ϕ = 100, ∇f = 0 (1, re-evaluation 2).  Armijo: 1 − 100 ≥ 0.25 ✗
#  → shrink: a = 0.125: x = 0.75 → 100 (3) ✗ → return (ϕ = 100, a = 0.25, 3, :success), xp = (0.75, 0), df_xp = 0.
#  optim.jl: f = 100, ‖g‖ = 0 < ϵ at the next loop top and f > f(x0) = 1 → :increasing_objective, iters_ran = 1.
KATS["E-increasing-objective"] = dict(
    case=closure_case("kat-increasing", plateau, [1.0, 0.0], beta="DaiYuan", eps=1e-5, max_iters=10,
                      ls="Backtracking", c1=0.25, discount=0.5, ls_max_iters=100, feas_max_iters=50),
    status="increasing_objective", iters=1, objective=100.0, minimizer=[0.75, 0.0], gradient=[0.0, 0.0],
    trace_f=[100.0], trace_g=[0.0], trace_a=[0.25], trace_e=[3], log_a=[0.25, 0.25, 0.125])

# ---------------------------------------------------------------------------------------------------------
# F. :non_finite_objective_or_gradient_proposed on minimizeobjective (optim.jl:108-121): StrongWolfeBisection accepts a
#  NaN objective — every comparison with NaN at nocedal.jl:81-105 is false and |dϕ| = 0 ≤ −c2·dϕ₀ (:107) holds.
#  nanpit, x0 = (1, 0): ϕ₀ = 1, u = (−2, 0), dϕ₀ = −4; a = 1: x = −1 → ϕ = NaN, ∇f = 0, dϕ = 0 → (NaN, 1, 1, :success) →
#  !isfinite(f_xp) → last good iterate x0, iters_ran = 0.
KATS["F-non-finite-proposed"] = dict(
    case=closure_case("kat-nonfinite", nanpit, [1.0, 0.0], beta="DaiYuan", eps=1e-9, max_iters=5, c1=1e-5, c2=0.8),
    status="non_finite_objective_or_gradient_proposed", iters=0, objective=1.0, minimizer=[1.0, 0.0], gradient=[2.0, 0.0],
    trace_f=[], trace_g=[], trace_a=[], trace_e=[], log_a=[1.0])


def check(out, k, name):
    assert out.status == k["status"], (name, out.status)
    assert out.iters_ran == k["iters"], (name, out.iters_ran)
    assert list(out.log_a) == k["log_a"], (name, list(out.log_a)[:8], len(out.log_a))
    assert out.objective == k["objective"], (name, out.objective)
    assert list(out.minimizer) == k["minimizer"] and list(out.gradient) == k["gradient"], (name, out.minimizer, out.gradient)
    assert list(out.trace_objective) == k["trace_f"] and list(out.trace_grad_norm) == k["trace_g"], name
    assert list(out.trace_step_size) == k["trace_a"] and list(out.trace_objective_evals) == k["trace_e"], name


@pytest.mark.parametrize("name", sorted(KATS))
def test_kat_oracle_c(name):
    check(run_oracle(KATS[name]["case"]), KATS[name], name)


@pytest.mark.parametrize("name", sorted(KATS))
def test_kat_oracle_numpy(name):
    check(run_numpy(KATS[name]["case"]), KATS[name], name)


@pytest.mark.parametrize("name", sorted(KATS))
def test_kat_product_host_engine(cgo, name):
    """csrc/cgo_engine.cpp + cgo_ctl.hpp over the test double, 1- and 3-point launches, with and without the
    emulated on-device controller."""
    for pts, depth in ((1, 0), (3, 0), (3, 4), (7, 0)):
        check(run_hostsim(KATS[name]["case"], points=pts, ctl_depth=depth), KATS[name], f"{name} pts={pts} ctl={depth}")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(KATS))
def test_kat_product_gpu(cgo, gpu_ctx, name):
    """libcgo_hip.so: the closures run through cgo_objective_create_callback — device-resident x, u, g, g⁺, device
    AXPY / direction / dot kernels, host objective."""
    from _cases import run_gpu
    check(run_gpu(KATS[name]["case"]), KATS[name], name)


def test_status_symbols_without_a_case_are_unreachable():
    """Two status symbols of SURVEY.md §5 have no case because the reference cannot produce them:

    :linesearch_a_max_overflow (nocedal.jl:143-149) needs `a > a·growth` with growth > 1 asserted (nocedal.jl:26) and
    a > 0 finite (a starts at a sanitised a_initial > 0, nocedal.jl:49-52, and only grows by (a·growth + a)/2): for
    finite a the product is ≥ a, and once it overflows, a·growth = Inf > a is false the other way round; a = Inf gives
    Inf > Inf = false.  Dead code, as SURVEY.md §3.2 notes.

    :bisection_lower_bound_larger_than_proposed_step (wolfe.jl:186-188) needs lb > a on entry of findfeasiblestepsize!.
    Its callers pass (a, lb) = (a_initial > 0, 0) (wolfe.jl:51), (2a or (lb+ub)/2 with lb < ub, lb) after :81-116 — both
    ≥ lb — or, after a collapsed bracket, either (a_initial, 0) (:125-129) or a ∈ {lb, ub} (:131 fall-through), and
    geometric.jl:60 passes lb = 0 with a = |ϕ₀|/u·u ≥ 0 or a previous step > 0.  NaN never satisfies lb > a.  Even if it
    fired, the flag is internal: the caller turns every flag ≠ :feasible into :cannot_find_(initial_)feasible_step.
    The state machines keep both branches (cgo_ctl.hpp); this test pins the arithmetic facts the argument rests on."""
    for a in (5e-324, 1e-300, 1.0, 1e300, 1.7976931348623157e308, float("inf")):
        for growth in (1.0000000000000002, 2.0, 1e10):
            assert not (a > a * growth)
    assert not (float("nan") > 1.0) and not (1.0 > float("nan"))
    lb, ub = 1.0, 1.0 + 2.0 ** -52
    assert (lb + ub) / 2 in (lb, ub)           # a collapsed bracket yields a ∈ {lb, ub}: never below lb


def test_ywl_inequalities_both_branches_of_each_min(cgo):
    """YuanWeiLuWolfe (wolfe.jl:240-248):  ϕa ≤ ϕ0 + c1·a·dϕ0 + a·min(−δ1·dϕ0, c1·a·‖u‖²/2)  and
    dϕa ≥ c2·dϕ0 + min(−δ1·dϕ0, c1·a·‖u‖²).  c1 = 1/4, c2 = 1/2, δ1 = 1/8, ϕ0 = 10, dϕ0 = −8, ‖u‖² = 16: −δ1·dϕ0 = 1,
    c1·a·‖u‖²/2 = 2a, c1·a·‖u‖² = 4a.
      a = 1/8:  min(1, 1/4) = 1/4 → rhs1 = 10 − 1/4 + 1/32 = 9.78125;   min(1, 1/2) = 1/2 → rhs2 = −4 + 1/2 = −3.5
      a = 1/4:  min(1, 1/2) = 1/2 → rhs1 = 10 − 1/2 + 1/8  = 9.625;     min(1, 1)   = 1   → rhs2 = −3
      a = 1:    min(1, 2)   = 1   → rhs1 = 10 − 2 + 1      = 9;         min(1, 4)   = 1   → rhs2 = −3
    Both inequalities are inclusive."""
    from oracle import oracle as O
    from oracle import cgo_oracle_np as N
    u = np.array([4.0, 0.0])
    y_p, y_o, y_n = cgo.YuanWeiLuWolfe(0.25, 0.5, 0.125), O.wolfe_bisection("YuanWeiLuWolfe", 0.25, 0.5, 0.125), N.YuanWeiLuWolfe(0.25, 0.5, 0.125)
    eps = 2.0 ** -40
    for a, rhs1, rhs2 in ((0.125, 9.78125, -3.5), (0.25, 9.625, -3.0), (1.0, 9.0, -3.0)):
        for phi, dphi, want in ((rhs1, rhs2, (True, True)), (rhs1 + eps, rhs2, (False, True)), (rhs1, rhs2 - eps, (True, False)),
                                (rhs1 - eps, rhs2 + eps, (True, True))):
            assert cgo.evalwolfeconditions(y_p, phi, dphi, a, u, 10.0, -8.0) == want, (a, phi, dphi)
            assert O.evalwolfeconditions(y_o, phi, dphi, a, u, 10.0, -8.0) == want, (a, phi, dphi)
            assert tuple(bool(v) for v in N.evalwolfeconditions(y_n, phi, dphi, a, u, 10.0, -8.0)) == want, (a, phi, dphi)
