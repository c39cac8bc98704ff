"""cgo_solver_policy (include/cgo.h): every way a solve can be RUN is selectable through the C ABI — per solver
(cgo_solver_create_ex), per context (cgo_ctx_set_default_policy) — and an explicit field beats the CGO_* experiment
overrides of earlier rounds.  A policy never changes WHAT is computed: every selection below is
held bit for bit to the same solve selected through the environment variable it replaces, and the parity suites hold
those to the oracle.
"""
import ctypes as C

import numpy as np
import pytest

from _cases import BIGN, Case, Out, _product_structs, first_divergence, gpu_objective, quad_D, rel
from _suite import rosen_x0


def test_policy_struct_layout_and_defaults(cgo):
    """CPU tier: the ctypes mirror has the C struct's size, and cgo_solver_policy_init writes "library policy" everywhere."""
    from cgo_amd import _lib
    c = _lib.SolverPolicyC()
    _lib.lib().cgo_solver_policy_init(C.byref(c))
    assert c.size == C.sizeof(_lib.SolverPolicyC) == 120
    assert (c.points, c.stored_gradient, c.placement_stages, c.placement_max_bytes, c.lbfgs_form, c.resident_points, c.resident_chunk) == (0,) * 7
    assert (c.resident, c.controller_depth, c.controller_graph, c.controller_fused, c.fused_tail, c.strict_tail, c.placement_search,
            c.lbfgs_fuse_grad, c.lbfgs_fuse_trial, c.lse_fixed_reference) == (-1,) * 10
    assert c.hbm_stream_bytes == 0.0 and list(c.reserved) == [0] * 8
    p = cgo.SolverPolicy(points=5, resident=False, lbfgs_form="gram", hbm_stream_bytes=1.0, placement_max_bytes=1 << 33)._c()
    assert (p.points, p.resident, p.lbfgs_form, p.hbm_stream_bytes, p.placement_max_bytes, p.strict_tail) == (5, 0, 3, 1.0, 1 << 33, -1)


def run(cgo, c: Case, policy=None, ctx=None):
    _, _lib, cfg, ls = _product_structs(c)
    obj = gpu_objective(c, ctx)
    s = cgo.Solver(obj, cfg, ls, policy)
    try:
        s.enable_trial_log()
        s.set_x0(c.x0)
        s.profile(True)
        s.start()
        while not s.iterate(1 << 40):
            pass
        r = s.results()
        la, lp, ld = s.trial_log()
        facts = dict(policy=s.policy(), family=s.kernel_family(), kinds=sorted(s.profile_get()), ctl=s.controller_launches(),
                     pushes=s.lbfgs_stats(), resident=s.resident_stats(), sym=s.kernel_symbol("accept_dir_trial"),
                     placement=s.placement_info())
    finally:
        s.close(); obj.close()
    return Out(r.objective, r.minimizer, r.gradient, r.iters_ran, r.status, r.trace.objective, r.trace.grad_norm, r.trace.step_size,
               r.trace.objective_evals, la, lp, ld, r.total_fdf_evals, r.total_launches), facts


def same(a: Out, b: Out):
    return (first_divergence(a, b) is None and np.array_equal(a.minimizer, b.minimizer) and a.objective == b.objective
            and a.total_launches == b.total_launches and np.array_equal(a.trace_grad_norm, b.trace_grad_norm))


def _quad(n=100003, iters=10, **kw):
    return Case("pol-quad", "quad_diag", n, np.ones(n), beta="PolakRibiere", D=quad_D(n), eps=1e-200, max_iters=iters, c2=0.1, **kw)


def _lse(n=60001, iters=10):
    from oracle import oracle as O
    return Case("pol-lse", "lse", n, 5.0 * O.fill_uniform(n, 24, -1.0, 1.0), beta="LBFGS", m=5, lam=1e-6, eps=1e-200, max_iters=iters, c2=0.9)


SELECTIONS = [   # (name, case factory, policy kwargs, the environment override that selected the same thing before)
    ("points1", _quad, dict(points=1, resident=False), {"CGO_MULTI_MIN_N": BIGN, "CGO_MULTI5_MIN_N": BIGN, "CGO_MULTI7_MIN_N": BIGN, "CGO_RESIDENT": "0"}),
    ("points3", _quad, dict(points=3, resident=False), {"CGO_MULTI_MIN_N": "0", "CGO_MULTI5_MIN_N": BIGN, "CGO_MULTI7_MIN_N": BIGN, "CGO_RESIDENT": "0"}),
    ("points5", _quad, dict(points=5, resident=False), {"CGO_MULTI_MIN_N": "0", "CGO_MULTI5_MIN_N": "0", "CGO_MULTI7_MIN_N": BIGN, "CGO_RESIDENT": "0"}),
    ("host-driven", _quad, dict(resident=False), {"CGO_RESIDENT": "0"}),
    ("resident-1pt-chunks", lambda: _quad(20000), dict(resident=True, resident_points=1, resident_chunk=1024), {"CGO_RESIDENT": "1", "CGO_RES_POINTS": "1", "CGO_RES_CHUNK": "1024"}),
    ("controller", lambda: _quad(20000), dict(resident=False, points=3, controller_depth=8),
     {"CGO_RESIDENT": "0", "CGO_MULTI_MIN_N": "0", "CGO_MULTI5_MIN_N": BIGN, "CGO_MULTI7_MIN_N": BIGN, "CGO_CTL_DEPTH": "8"}),
    ("controller-unfused", lambda: _quad(20000), dict(resident=False, points=3, controller_depth=8, controller_fused=False),
     {"CGO_RESIDENT": "0", "CGO_MULTI_MIN_N": "0", "CGO_MULTI5_MIN_N": BIGN, "CGO_MULTI7_MIN_N": BIGN, "CGO_CTL_DEPTH": "8", "CGO_CTL_FUSED": "0"}),
    ("stored-gradient", _quad, dict(resident=False, stored_gradient=True), {"CGO_RESIDENT": "0", "CGO_STORED_G": "1"}),
    ("pure-hbm-everywhere", _quad, dict(resident=False, hbm_stream_bytes=1.0), {"CGO_RESIDENT": "0", "CGO_BIG_BYTES": "1"}),
    ("finalize-launches", _quad, dict(resident=False, fused_tail=False), None),
    ("strict-tail", _quad, dict(resident=False, strict_tail=True), None),
    ("lbfgs-one-pass-own-update", _lse, dict(lbfgs_form="one_pass_own_update"), {"CGO_LBFGS_SPEC": "1"}),
    ("lbfgs-gram", _lse, dict(lbfgs_form="gram"), {"CGO_LBFGS_SPEC": "0"}),
    ("lbfgs-gram-plain-push", _lse, dict(lbfgs_form="gram", lbfgs_fuse_grad=False), {"CGO_LBFGS_SPEC": "0", "CGO_LBFGS_FUSE_GRAD": "0"}),
    ("lbfgs-gram-own-trial", _lse, dict(lbfgs_form="gram", lbfgs_fuse_trial=False), {"CGO_LBFGS_SPEC": "0", "CGO_LBFGS_FUSE_TRIAL": "0"}),
    ("lbfgs-two-loop", _lse, dict(lbfgs_form="two_loop"), {"CGO_LBFGS_TWO_LOOP": "1"}),
    ("lse-running-max", _lse, dict(lse_fixed_reference=False), {"CGO_LSE_REF": "0"}),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,case,kw,env", SELECTIONS, ids=[s[0] for s in SELECTIONS])
def test_policy_selects_what_the_environment_override_selected(cgo, gpu_ctx, monkeypatch, name, case, kw, env):
    for k in ("CGO_RESIDENT", "CGO_MULTI_MIN_N", "CGO_MULTI5_MIN_N", "CGO_MULTI7_MIN_N", "CGO_CTL_DEPTH", "CGO_LBFGS_SPEC"):
        monkeypatch.delenv(k, raising=False)
    c = case()
    ctx = cgo.Context(0)           # its own context: fused_tail / strict_tail are context-wide
    try:
        got, facts = run(cgo, c, cgo.SolverPolicy(**kw), ctx)
        pol = facts["policy"]
        for k, v in kw.items():    # what the solver says it runs with
            want = cgo.LBFGS_FORMS[v] if k == "lbfgs_form" else (int(v) if isinstance(v, bool) else v)
            if k == "resident" and v and not facts["resident"][0]:
                continue
            assert pol[k] == want, (k, pol[k], want)
        if "points" in kw:
            assert f", {kw['points']}, " in facts["sym"] or kw.get("stored_gradient"), facts["sym"]
        if kw.get("resident") is False:
            assert facts["resident"] == (0, 0) and "resident" not in facts["kinds"]
        if kw.get("resident"):
            assert facts["resident"][1] > 0
        if kw.get("controller_depth"):
            assert facts["ctl"] > 0
        if kw.get("stored_gradient"):
            assert facts["family"].startswith("k_fused")
        if kw.get("hbm_stream_bytes") == 1.0:
            assert facts["sym"].endswith("true>"), facts["sym"]
        if kw.get("lbfgs_form") == "gram":
            assert facts["pushes"][0] == 0 and facts["pushes"][1] + facts["pushes"][2] > 0
        if kw.get("lbfgs_form") == "one_pass_own_update":
            assert facts["pushes"][0] > 0
    finally:
        ctx.close()
    # the plain default on a fresh context: the same solve (a policy never changes the step sequence)
    ctx = cgo.Context(0)
    try:
        base, _ = run(cgo, c, None, ctx)
        assert first_divergence(got, base) is None and rel(got.minimizer, base.minimizer) <= 1e-11
        if env is not None:        # and exactly the run the environment override used to select
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            old, _ = run(cgo, c, None, ctx)
            assert same(got, old), name
    finally:
        ctx.close()


@pytest.mark.gpu
def test_an_explicit_policy_beats_the_environment_and_the_context_default(cgo, gpu_ctx, monkeypatch):
    c = _quad(20000)
    monkeypatch.setenv("CGO_RESIDENT", "1")
    ctx = cgo.Context(0)
    try:
        _, f_env = run(cgo, c, None, ctx)
        assert f_env["resident"][1] > 0                                   # the override applies where nobody chose
        _, f_arg = run(cgo, c, cgo.SolverPolicy(resident=False), ctx)
        assert f_arg["resident"] == (0, 0)                                # explicit argument > environment
        ctx.set_default_policy(cgo.SolverPolicy(resident=False, points=3))
        _, f_def = run(cgo, c, None, ctx)
        assert f_def["resident"] == (0, 0) and ", 3, " in f_def["sym"]    # context default > environment
        _, f_both = run(cgo, c, cgo.SolverPolicy(points=7), ctx)
        assert f_both["resident"] == (0, 0) and ", 7, " in f_both["sym"]  # argument and default merge field by field
        r = cgo.minimizeobjective(gpu_objective(c, ctx), c.x0, _product_structs(c)[2], _product_structs(c)[3])
        assert r.iters_ran == c.max_iters                                 # the one-shot entry points inherit the default
        ctx.set_default_policy(None)
        _, f_reset = run(cgo, c, None, ctx)
        assert f_reset["resident"][1] > 0
    finally:
        ctx.close()


@pytest.mark.gpu
def test_invalid_policies_are_refused(cgo, gpu_ctx):
    c = _quad(4096)
    for bad in (dict(points=4), dict(resident_points=5), dict(placement_stages=9)):
        with pytest.raises(cgo.CgoError):
            run(cgo, c, cgo.SolverPolicy(**bad))
    from cgo_amd import _lib
    p = cgo.SolverPolicy()._c()
    p.size = 64                                            # a struct of another library version
    with pytest.raises(cgo.CgoError):
        _lib.check(_lib.lib().cgo_ctx_set_default_policy(gpu_ctx._h, C.byref(p)))


@pytest.mark.gpu
def test_placement_search_memory_cap(cgo, gpu_ctx, monkeypatch):
    """The search's transient memory is an argument, not a surprise: capped below one stage of spares it does not run, capped
    at one stage it times candidates from eight spares only; the solve is the same either way."""
    monkeypatch.delenv("CGO_PLACE_TUNE", raising=False)
    n = 40_000_000                                          # accept+dir+trial moves 1.6 GB: a pure-HBM launch
    c = Case("place", "quad_diag", n, np.ones(n), beta="PolakRibiere", D=quad_D(n), eps=1e-200, max_iters=3, c2=0.1)
    vec = 8 * n
    a, fa = run(cgo, c, cgo.SolverPolicy(placement_search=True, placement_max_bytes=4 * vec))
    b, fb = run(cgo, c, cgo.SolverPolicy(placement_search=True, placement_max_bytes=8 * vec, placement_stages=1))
    o, fo = run(cgo, c, None)                                # library policy: the search is opt-in
    assert fa["placement"][2] == 0 and fo["placement"][2] == 0           # (first_us, best_us, candidates)
    assert 1 <= fb["placement"][2] <= 64
    assert first_divergence(a, b) is None and first_divergence(a, o) is None and rel(a.minimizer, b.minimizer) <= 1e-12


@pytest.mark.gpu
def test_stencil_objective_launch_policy_is_fenced(cgo, gpu_ctx, monkeypatch):
    """include/cgo.h, CGO_OBJ_ROSENBROCK_CHAINED: whatever the policy asks for, the stencil objective runs host-driven launches of
    one or three trial points above ≈ 4 800 elements (no 5/7-point launches, no controller, no multi-workgroup resident form),
    and one resident workgroup below — pinned by kernel symbols and launch counts (finish or fence)."""
    for k in ("CGO_RESIDENT", "CGO_MULTI_MIN_N", "CGO_MULTI5_MIN_N", "CGO_MULTI7_MIN_N", "CGO_CTL_DEPTH"):
        monkeypatch.delenv(k, raising=False)
    n = 100002
    big = Case("chain-big", "rosenbrock_chained", n, rosen_x0(n), beta="HagerZhang", max_iters=8, ls="WolfeBisection", cond="Wolfe",
               c1=1e-3, c2=0.9, ls_max_iters=100)
    base, f0 = run(cgo, big, None)
    assert f0["resident"] == (0, 0) and f0["ctl"] == 0 and f0["sym"] == "k_chain<7, 3, false>", f0
    greedy, f7 = run(cgo, big, cgo.SolverPolicy(points=7, controller_depth=8, resident=True))
    assert f7["sym"] == "k_chain<7, 3, false>" and f7["ctl"] == 0 and f7["resident"] == (0, 0)      # clamped to three points, host-driven
    assert same(greedy, base)
    one, f1 = run(cgo, big, cgo.SolverPolicy(points=1))
    assert f1["sym"] == "k_chain<7, 1, false>" and first_divergence(one, base) is None and one.total_launches >= base.total_launches
    with pytest.raises(cgo.CgoError):                                                                # CG β kinds only
        run(cgo, Case("chain-qn", "rosenbrock_chained", 1000, rosen_x0(1000), beta="LBFGS", m=4, max_iters=4), None)
    small = Case("chain-small", "rosenbrock_chained", 1000, rosen_x0(1000), beta="PolakRibiere", max_iters=6, c2=0.1)
    _, fs = run(cgo, small, None)
    assert fs["resident"][1] == 6 and fs["resident"][0] >= 1                                          # one resident workgroup: whole iterations in a launch
    mid = Case("chain-mid", "rosenbrock_chained", 6000, rosen_x0(6000), beta="PolakRibiere", max_iters=6, c2=0.1)
    _, fm = run(cgo, mid, cgo.SolverPolicy(resident=True))
    assert fm["resident"] == (0, 0)                                                                   # above one workgroup's LDS: launches
