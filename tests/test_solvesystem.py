"""CPU tier for solvesystem (reference src/engine/solve_system.jl:64-253, exported at
ConjugateGradientOptim.jl:28): the C oracle, the independently written numpy restatement and the
PRODUCT's host engine (over the test double) must walk the same trajectories; hand-derived known
answers pin the first two iterations — including the reference's re-basing bug."""
import json
import os

import numpy as np
import pytest

from _cases import Case, O, N, assert_parity, first_divergence, rel, run_hostsim, run_numpy, run_oracle
from _suite import sys_cases, sys_status_cases

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))["solvesystem_first_iterations"]


def _kat_case(max_iters):
    k = KAT
    return Case("sys-kat", "quad_diag", 2, np.array(k["x0"]), beta="HagerZhang", D=np.array(k["D"]), eps=1e-12,
                max_iters=max_iters, ls="SolveSys", sys_s=k["s"], sys_sigma=k["sigma"], sys_rho=k["rho"])


@pytest.mark.parametrize("runner", [run_oracle, run_numpy, run_hostsim], ids=["c-oracle", "numpy-oracle", "engine"])
def test_hand_derived_first_iterations(cgo, runner):
    """Identity system g(x) = x from x0 = (3, 4), s = 0.25, σ = 0.5, ρ = 0.5, HagerZhang — every value is
    exactly representable (derivation in tests/golden/make_golden.py).  Iteration 2 lands on
    x0 + m₂·g(z₂) = (1.875, 2.5), NOT on x₁ + m₂·g(z₂) = (1.125, 1.5): updateiteratesolvesys! is applied to
    the `x_next` buffer, which then still holds x0 (solve_system.jl:172-178,194)."""
    k = KAT
    r1 = runner(_kat_case(1))
    assert r1.status == "max_iters_reached" and r1.iters_ran == 1
    assert np.array_equal(r1.minimizer, k["x1"]) and r1.objective == k["f1"]
    assert list(r1.trace_grad_norm) == [k["norm_g1"]] and list(r1.trace_step_size) == [k["a1"]]
    assert list(r1.trace_objective_evals) == [k["evals_index1"]]     # the 0-based index of the accepted trial (:52)
    r2 = runner(_kat_case(2))
    assert np.array_equal(r2.minimizer, k["x2_reference"]) and not np.array_equal(r2.minimizer, k["x2_if_rebased"])
    assert list(r2.trace_step_size) == [k["a1"], k["a2"]]
    assert list(r2.log_a) == k["trial_steps"]


def test_hand_derived_beta_and_direction(cgo):
    """β_HZ = 3/4 and u₁ = (−4.5, −6) after the first projection step: visible through the second line
    search's first trial, ϕ-independent: dϕ(a) = g(x₁ + a·u₁)·u₁ = −(1 − 2a)·14.0625·… (exact values in kat.json)."""
    r = run_oracle(_kat_case(2))
    assert r.log_dphi[1] == KAT["dphi_second_search_first_trial"]


@pytest.mark.parametrize("c", sys_cases(small_only=True), ids=lambda c: c.name)
def test_oracles_agree(cgo, c):
    assert_parity(run_numpy(c), run_oracle(c), 1e-10, c.name)


@pytest.mark.parametrize("c", sys_cases(small_only=True), ids=lambda c: c.name)
def test_engine_matches_oracle(cgo, c):
    got, ref = run_hostsim(c), run_oracle(c)
    assert_parity(got, ref, 1e-10, c.name)
    assert got.total_fdf_evals == ref.total_fdf_evals
    assert np.allclose(got.trace_grad_norm, ref.trace_grad_norm, rtol=1e-9, atol=0)


@pytest.mark.parametrize("c", sys_cases(sizes=(64,), small_only=True), ids=lambda c: c.name)
def test_engine_speculation_and_slicing_change_nothing(cgo, c):
    """Three steps s·ρ^i per launch vs one; iterate() in slices: same trajectory, fewer launches."""
    multi, single = run_hostsim(c), run_hostsim(c, chunk=-1)
    assert first_divergence(multi, single) is None and np.array_equal(multi.minimizer, single.minimizer)
    assert multi.total_fdf_evals == single.total_fdf_evals and multi.total_launches < single.total_launches
    for chunk in (1, 3):
        part = run_hostsim(c, chunk=chunk)
        assert first_divergence(part, multi) is None and np.array_equal(part.minimizer, multi.minimizer)
        assert part.status == multi.status and part.iters_ran == multi.iters_ran


@pytest.mark.parametrize("want,iters,c", sys_status_cases(), ids=lambda v: v.name if isinstance(v, Case) else None)
def test_status_paths(cgo, want, iters, c):
    ref, alt, got = run_oracle(c), run_numpy(c), run_hostsim(c)
    for r in (alt, got):
        assert r.status == ref.status and r.iters_ran == ref.iters_ran
        assert len(r.trace_objective) == r.iters_ran
        if np.all(np.isfinite(ref.minimizer)):
            assert rel(r.minimizer, ref.minimizer) <= 1e-10
    assert got.total_fdf_evals == ref.total_fdf_evals
    if want is not None:
        assert ref.status == want
    if iters is not None:
        assert ref.iters_ran == iters


def test_early_exit_returns_the_trial_point(cgo):
    """solve_system.jl:145-166: when the accepted TRIAL point already satisfies ‖g‖ < ϵ the result is
    (xp, g(xp), f(xp)) — not an iterate of the projection recursion — and the trace gets one more row."""
    c = [c for _, _, c in sys_status_cases() if c.name == "sys-st-trial-point"][0]
    for r in (run_oracle(c), run_hostsim(c)):
        assert r.status == "success" and r.iters_ran == 1
        assert np.array_equal(r.minimizer, np.zeros(c.n)) and np.array_equal(r.gradient, np.zeros(c.n)) and r.objective == 0.0
        assert list(r.trace_step_size) == [1.0] and list(r.trace_objective_evals) == [0]


def test_setup_asserts_and_default_max_iters(cgo):
    """setupLinesearchSolveSys (solve_system.jl:13-27): ρ ∈ (0,1), s > 0 asserted; σ never is;
    max_iters defaults to round(Int, log(ρ, 1e-6))."""
    import cgo_amd
    assert cgo_amd.setupLinesearchSolveSys(1.0).max_iters == 269                  # log(1e-6)/log(0.95) = 269.35
    assert cgo_amd.setupLinesearchSolveSys(1.0, ρ=0.5).max_iters == 20            # 19.93
    assert O.linesearch_solve_sys(1.0).max_iters == 269 and N.LinesearchSolveSys(1.0).max_iters == 269
    assert cgo_amd.setupLinesearchSolveSys(1.0, σ=-3.0).σ == -3.0                 # not checked by the reference either
    for bad in (dict(s=0.0), dict(s=1.0, ρ=1.0), dict(s=1.0, ρ=0.0), dict(s=-1.0)):
        with pytest.raises(AssertionError):
            cgo_amd.setupLinesearchSolveSys(**bad)


def test_reference_throws_switch(cgo):
    """Where the reference throws UndefVarError (solve_system.jl:55) the shim can do the same."""
    import cgo_amd
    assert issubclass(cgo_amd.UndefVarError, RuntimeError)
    with pytest.raises(TypeError):
        cgo_amd.solvesystem(object(), [1.0], cgo_amd.setupCGConfig(1e-5, cgo_amd.HagerZhang(), cgo_amd.EnableTrace()),
                            cgo_amd.setupStrongWolfeBisection(1e-5, 0.8))
