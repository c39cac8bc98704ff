"""The parity case list shared by the CPU tier (oracle ↔ numpy oracle ↔ host engine
over the test double) and the GPU tier (libcgo_hip.so ↔ oracle, golden fixtures).

Horizons are kept short enough that the reference's own reduction-order noise
(C oracle vs numpy oracle = two valid summation orders of the same formulas)
stays below the 1e-10 bar; HZ/YWS lose ≈ 0.25 digits per iteration to
cancellation in getβ (cg_flavours.jl:73-76,102-105), so the bar cannot hold on
long HZ runs for ANY two implementations (see tests/test_oracle.py::test_noise_floor).
"""
from __future__ import annotations

import numpy as np

from _cases import Case, quad_D, O

BETAS = ["HagerZhang", "YuanWangSheng", "SallehAlhawarat", "LiuStorrey", "PolakRibiere",
         "HestenesStiefel", "DaiYuan"]


def rosen_x0(n, jitter=0.01, seed=7):
    return np.tile([-1.2, 1.0], n // 2) + jitter * O.fill_uniform(n, seed, -1.0, 1.0)


def parity_cases(sizes=(31, 64, 1000, 100003), small_only=False):
    cs = []
    # examples/min.jl: Booth, HagerZhang, StrongWolfeBisection(1e-5, 0.8), ϵ=1e-5, x0=[0.43,1.23]
    cs.append(Case("booth-min.jl", "booth", 2, np.array([0.43, 1.23]), beta="HagerZhang"))
    for b in BETAS:
        cs.append(Case(f"booth-{b}", "booth", 2, np.array([0.43, 1.23]), beta=b, max_iters=60,
                       c2=0.1 if b == "PolakRibiere" else 0.8))
    for n in sizes:
        if small_only and n > 2000:
            continue
        D = quad_D(n)
        x0 = np.ones(n)
        for b in BETAS:
            cs.append(Case(f"quad{n}-{b}-SW", "quad_diag", n, x0, beta=b, D=D, eps=1e-9, max_iters=16,
                           c2=0.1 if b == "PolakRibiere" else 0.8))
        cs.append(Case(f"quad{n}-HZ-Wolfe", "quad_diag", n, x0, beta="HagerZhang", D=D, eps=1e-9,
                       max_iters=16, ls="WolfeBisection", cond="Wolfe", c1=1e-3, c2=0.9,
                       ls_max_iters=100))
        cs.append(Case(f"quad{n}-DY-YWL", "quad_diag", n, x0, beta="DaiYuan", D=D, eps=1e-9,
                       max_iters=16, ls="WolfeBisection", cond="YuanWeiLuWolfe", c1=1e-3, c2=0.9,
                       delta1=1e-4, ls_max_iters=100))
    for n in (2, 32, 1000):
        x0 = rosen_x0(n)
        cs.append(Case(f"rosen{n}-HZ-Wolfe", "rosenbrock_paired", n, x0, beta="HagerZhang",
                       max_iters=12, ls="WolfeBisection", cond="Wolfe", c1=1e-3, c2=0.9, ls_max_iters=100))
        cs.append(Case(f"rosen{n}-DY-SW", "rosenbrock_paired", n, x0, beta="DaiYuan", max_iters=12, c2=0.8))
        cs.append(Case(f"rosen{n}-SA-SW", "rosenbrock_paired", n, x0, beta="SallehAlhawarat", max_iters=12, c2=0.8))
    # BASELINE config 1, literally: (paired) Rosenbrock n = 1000 + Polak–Ribière + StrongWolfeBisection with the
    # examples/min.jl plumbing (c1 = 1e-5, growth 2, 1000/100 iterations, ϵ = 1e-5) and x0 = (−1.2, 1, …) exactly.
    # c2 = 0.1: with min.jl's 0.8 plain PR leaves the descent cone at iteration 2 (st-nondescent).  Five iterations (47
    # evaluations): from this x0 PR amplifies reduction-order noise tenfold every two iterations — the two oracles
    # already differ by 2e-11 after 6 iterations and 7e-10 after 10 — so the 1e-10 bar is only meaningful this far.
    cs.append(Case("rosen1000-PolakRibiere-SW", "rosenbrock_paired", 1000, np.tile([-1.2, 1.0], 500), beta="PolakRibiere",
                   max_iters=5, c2=0.1, extra={"x0": ["rosen", 0.0, 7]}))
    return cs


def reset_cases():
    """WolfeBisection's bracket collapse on a direction u ≠ −g (wolfe.jl:122-130): the search restarts from
    steepest descent with the OLD dϕ₀, and the getβ that follows sees the RESET u — SallehAlhawarat's
    denominator dot(u, g⁺) − dot(u, g) (cg_flavours.jl:145) is the flavour that reads it.  The collapse only
    happens once the iterates sit at rounding level, i.e. late (iteration 146 of 160 here); n = 2 keeps every
    sum a single pair, so all implementations add in the same order and the long horizon holds bit for bit.
    (Round 1's engine kept the stale dϕ₀ for getβ and left this trajectory at iteration 145.)"""
    kw = dict(eps=1e-300, max_iters=200, ls="WolfeBisection", cond="Wolfe", c1=1e-3, c2=0.9, ls_max_iters=400)
    return [Case("reset-rosen2-SA", "rosenbrock_paired", 2, rosen_x0(2, 0.3, 3), beta="SallehAlhawarat", **kw),
            Case("reset-rosen2-DY", "rosenbrock_paired", 2, rosen_x0(2, 0.5, 11), beta="DaiYuan", **kw)]


def backtracking_cases():
    """Backtracking/Armijo (geometric.jl) — bug-for-bug parity incl. the adopted rejected trial."""
    n = 1000
    D = quad_D(n)
    bt = dict(ls="Backtracking", c1=1e-3, discount=0.5, ls_max_iters=100)
    return [
        Case("bt-quad-DY", "quad_diag", n, np.ones(n), beta="DaiYuan", D=D, eps=1e-9, max_iters=16, **bt),
        Case("bt-quad-HZ", "quad_diag", n, np.ones(n), beta="HagerZhang", D=D, eps=1e-9, max_iters=16, **bt),
        Case("bt-rosen-HZ", "rosenbrock_paired", n, rosen_x0(n), beta="HagerZhang", max_iters=12, **bt),
        Case("bt-rosen-LBFGS", "rosenbrock_paired", 64, rosen_x0(64), beta="LBFGS", m=5, max_iters=12, **bt),
        Case("bt-booth-HZ", "booth", 2, np.array([0.43, 1.23]), beta="HagerZhang", max_iters=40,
             ls="Backtracking", c1=1e-3, discount=0.7, ls_max_iters=100),
        Case("bt-quad-grow", "quad_diag", n, np.ones(n) * 1e-3, beta="DaiYuan", D=D * 1e-4, eps=1e-12, max_iters=10, **bt),
        Case("bt-quad31-SA", "quad_diag", 31, np.ones(31), beta="SallehAlhawarat", D=quad_D(31), eps=1e-9, max_iters=16, **bt),
    ]


def status_cases():
    """One case per reachable status symbol of the path."""
    n = 64
    D = quad_D(n)
    x0 = np.ones(n)
    cs = []
    cs.append(("success", Case("st-success", "booth", 2, np.array([0.43, 1.23]))))
    cs.append(("max_iters_reached", Case("st-maxit", "quad_diag", n, x0, D=D, beta="DaiYuan", eps=1e-12, max_iters=3)))
    cs.append(("max_iters_reached", Case("st-maxit0", "quad_diag", n, x0, D=D, eps=1e-12, max_iters=0)))
    # plain PR + loose curvature condition → ascent direction at iteration 2 (nocedal.jl:57-63)
    cs.append(("non_descent_search_direction", Case("st-nondescent", "quad_diag", n, x0, D=D, beta="PolakRibiere", c2=0.8, max_iters=50)))
    cs.append(("non_descent_search_direction", Case("st-nondescent-w", "quad_diag", n, x0, D=D, beta="PolakRibiere",
                                                    ls="WolfeBisection", c1=1e-3, c2=0.9, ls_max_iters=100, max_iters=50)))
    cs.append(("zoom_max_iters_reached", Case("st-zoom", "quad_diag", n, x0, D=D, zoom_max_iters=2, max_iters=50)))
    cs.append(("linesearch_max_iters_reached", Case("st-lsmax", "quad_diag", n, x0, D=D * 1e-6, eps=1e-12, ls_max_iters=2, c2=0.1, max_iters=50)))
    cs.append(("linesearch_max_iters_reached", Case("st-lsmax-w", "quad_diag", n, x0, D=D, ls="WolfeBisection", c1=1e-3, c2=0.9,
                                                    ls_max_iters=2, max_iters=50)))
    cs.append(("max_step_length_reached", Case("st-maxstep", "quad_diag", n, x0, D=D * 1e-6, ls="WolfeBisection", c1=1e-3, c2=0.9,
                                               ls_max_iters=100, max_step_size=4.0, max_iters=50)))
    # x0 already optimal → :success with 0 iterations (optim.jl:53-66)
    cs.append(("success", Case("st-x0-optimal", "booth", 2, np.array([1.0, 3.0]))))
    # overflowing objective: f(x0) finite, every trial overflows to Inf/NaN
    big = np.full(n, 1e200)
    # overflow to Inf inside the line search: whatever the reference's state machine does, do the same
    cs.append((None, Case("st-overflow", "quad_diag", n, x0, D=big, zoom_max_iters=3, ls_max_iters=1, max_iters=5)))
    cs.append(("accepted_non_finite_iterate", Case("st-nonfinite-x0", "quad_diag", n, np.full(n, 1e200), D=big, ls="WolfeBisection", c1=1e-3, c2=0.9,
                                                   ls_max_iters=10, max_iters=5)))
    cs.append(("cannot_find_initial_feasible_step", Case("st-infeasible0", "quad_diag", n, np.full(n, 1e150), D=np.full(n, 1e3), ls="WolfeBisection",
                                                         c1=1e-3, c2=0.9, ls_max_iters=10, feas_max_iters=2, max_iters=5)))
    # LinearAlgebra.norm is the TRUE 2-norm (BLAS.nrm2 / generic_norm2): finite although Σg² overflows …
    cs.append((None, Case("st-norm-overflow", "quad_diag", n, x0, D=D, ls="Backtracking", c1=1e-3, discount=1e-300,
                          ls_max_iters=10, max_iters=5)))
    # … and non-zero although every g_i² underflows (‖g‖ ≈ 1e-166 > ϵ = 1e-200: no spurious convergence)
    cs.append((None, Case("st-norm-underflow", "quad_diag", n, x0 * 1e-170, D=D, eps=1e-200, beta="DaiYuan", max_iters=3)))
    # getβ's own norms are LinearAlgebra.norm too: SallehAlhawarat squares norm(g⁺) (cg_flavours.jl:140), YuanWangSheng
    # multiplies norm(u)·norm(y) (:65).  With |g_i| ≈ 1e-158…1e-155 every g_i² is subnormal (≈ 8 significant bits at
    # 1e-316), so sqrt(Σg²) is good to 1e-8 only while the scaled form is exact to rounding: a β built on the fast form
    # leaves the 1e-10 bar within two iterations (round 1 did).
    cs.append((None, Case("st-norm-underflow-sa", "quad_diag", n, x0 * 1e-158, D=D, eps=1e-300, beta="SallehAlhawarat", max_iters=4)))
    cs.append((None, Case("st-norm-underflow-yws", "quad_diag", n, x0 * 1e-158, D=D, eps=1e-300, beta="YuanWangSheng", max_iters=4)))
    # geometric.jl:127-133
    cs.append(("proposed_step_same_as_current_step", Case("st-bt-same", "quad_diag", n, x0, D=D, ls="Backtracking", c1=1e-3, discount=1.0,
                                                          ls_max_iters=10, max_iters=5)))
    cs.append(("non_finite_step_proposed", Case("st-bt-nonfinite", "quad_diag", n, x0, D=D, ls="Backtracking", c1=1e-3,
                                                discount=5e-324, ls_max_iters=10, max_iters=5)))
    cs.append(("linesearch_max_iters_reached", Case("st-bt-lsmax", "quad_diag", n, x0, D=D, ls="Backtracking", c1=1e-3, discount=0.5,
                                                    ls_max_iters=1, max_iters=5)))
    return cs


def sys_cases(sizes=(31, 64, 1000, 100003), small_only=False):
    """solvesystem (solve_system.jl:64-253).  The reference's projection step lands on the iterate of
    two iterations ago (x_next is never re-based, :172-178,:194), so its trajectories do not converge
    in general; parity is about reproducing them, statuses included.  Short horizons: the
    trajectories are chaotic, and reduction-order noise doubles every few iterations."""
    cs = []
    cs.append(Case("sys-booth-HZ", "booth", 2, np.array([0.43, 1.23]), beta="HagerZhang", eps=1e-6, max_iters=25,
                   ls="SolveSys", sys_s=0.1))
    for n in sizes:
        if small_only and n > 2000:
            continue
        x0 = np.ones(n)
        for b, s, hi in (("HagerZhang", 0.5, 2.0), ("PolakRibiere", 0.1, 10.0), ("LiuStorrey", 0.05, 1000.0),
                         ("YuanWangSheng", 0.5, 2.0), ("SallehAlhawarat", 0.1, 10.0), ("DaiYuan", 0.5, 2.0)):
            cs.append(Case(f"sys-quad{n}-{b}", "quad_diag", n, x0, beta=b, D=quad_D(n, 1.0, hi), eps=1e-9,
                           max_iters=8, ls="SolveSys", sys_s=s, extra={"D_lo": 1.0, "D_hi": hi}))
        if n % 2 == 0:
            cs.append(Case(f"sys-rosen{n}-HZ", "rosenbrock_paired", n, rosen_x0(n), beta="HagerZhang", eps=1e-9,
                           max_iters=8, ls="SolveSys", sys_s=0.01))
    return cs


def sys_status_cases():
    n = 64
    x0, D1 = np.ones(n), np.ones(n)
    cs = []
    # g(x0) already below ϵ → :success with 0 iterations (:112-123)
    cs.append(("success", 0, Case("sys-st-start", "quad_diag", n, 1e-3 * x0, D=D1, eps=0.5, max_iters=10, ls="SolveSys")))
    # identity system, s = 1: the first trial point is the root → early exit with the TRIAL point (:145-166)
    cs.append(("success", 1, Case("sys-st-trial-point", "quad_diag", n, x0, D=D1, eps=1e-8, max_iters=10, ls="SolveSys", sys_s=1.0)))
    # loose ϵ: the top-of-loop test after one projection step (:112)
    cs.append(("success", None, Case("sys-st-loop", "quad_diag", n, x0, D=D1, eps=0.9, max_iters=50, ls="SolveSys",
                                     sys_s=0.5, sys_rho=0.5)))
    cs.append(("max_iters_reached", 5, Case("sys-st-max", "quad_diag", n, x0, D=quad_D(n, 1.0, 2.0), eps=1e-12, max_iters=5,
                                            ls="SolveSys", sys_s=0.5)))
    # no step passes within max_iters trials: the reference throws here (:55); :linesearch_failed is its intent
    cs.append(("linesearch_failed", 0, Case("sys-st-lsfail", "quad_diag", n, x0, D=quad_D(n, 1.0, 1000.0), eps=1e-12, max_iters=5,
                                            ls="SolveSys", sys_s=1.0, sys_max_iters=3)))
    # f(x_next) overflows → last good iterate (:178-191)
    cs.append(("non_finite_objective_or_gradient_proposed", 0,
               Case("sys-st-nonfinite", "quad_diag", n, 1e200 * x0, D=D1, eps=1e-6, max_iters=5, ls="SolveSys", sys_s=0.5)))
    # Σg² ≈ 1e-288 is below the 1e-280 guard (still normal numbers): LinearAlgebra.norm's scaled path on every norm
    cs.append((None, None, Case("sys-st-norm-underflow", "quad_diag", n, 1e-145 * x0, D=quad_D(n, 1.0, 2.0), eps=1e-300, max_iters=3,
                                ls="SolveSys", sys_s=0.5, sys_rho=0.5)))
    cs.append(("max_iters_reached", 0, Case("sys-st-zero-iters", "quad_diag", n, x0, D=D1, eps=1e-6, max_iters=0, ls="SolveSys")))
    return cs


def broyden_cases():
    """BroydenFamily (qn_flavours.jl:53-90; mu carries θ).  The oracles run the reference's dense n×n algebra;
    the engine runs u = −g (B_new = B up to rounding, see include/cgo.h) — they must walk the same steps."""
    cs = []
    for theta in (1.0, 0.0):   # DFP (examples/constrained.jl:143) and BFGS
        cs.append(Case(f"bf-booth-theta{theta:g}", "booth", 2, np.array([0.43, 1.23]), beta="BroydenFamily", mu=theta, max_iters=60))
    for n in (31, 64):
        cs.append(Case(f"bf-quad{n}", "quad_diag", n, np.ones(n), beta="BroydenFamily", mu=1.0, D=quad_D(n), eps=1e-9, max_iters=16))
    cs.append(Case("bf-rosen32-wolfe", "rosenbrock_paired", 32, rosen_x0(32), beta="BroydenFamily", mu=1.0, eps=1e-9, max_iters=16,
                   ls="WolfeBisection", cond="Wolfe", c1=1e-3, c2=0.9))
    cs.append(Case("bf-booth-armijo", "booth", 2, np.array([0.43, 1.23]), beta="BroydenFamily", mu=1.0, max_iters=40,
                   ls="Backtracking", c1=1e-3, discount=0.9, ls_max_iters=300))
    return cs
