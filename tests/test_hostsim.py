"""CPU tier: the PRODUCT's host control plane (csrc/cgo_engine.cpp — outer loop, both
bisection line searches, β on reduced scalars, speculative first trial, trace/result
bookkeeping, rank-ordered cross-rank sums) driven over the test-double backend
(tests/hostsim/) and checked against the independent oracle.  The device kernels
are covered by the GPU tier; nothing here is a product code path on its own."""
import os
import subprocess
import sys

import numpy as np
import pytest

from _cases import Case, assert_parity, first_divergence, quad_D, rel, run_hostsim, run_oracle, sim_lib
from _suite import backtracking_cases, broyden_cases, parity_cases, reset_cases, status_cases, rosen_x0

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("c", parity_cases(small_only=True), ids=lambda c: c.name)
def test_engine_matches_oracle(cgo, c):
    """Default: speculative 3-point launches (requested step + the two possible next steps)."""
    assert_parity(run_hostsim(c), run_oracle(c), 1e-10, c.name)


@pytest.mark.parametrize("c", reset_cases(), ids=lambda c: c.name)
def test_engine_wolfe_reset_matches_oracle(cgo, c):
    """Through the steepest-descent restart of wolfe.jl:122-130 and the getβ after it (see reset_cases)."""
    for pts in (1, 3, 7):
        assert_parity(run_hostsim(c, points=pts), run_oracle(c), 1e-10, c.name)


def test_engine_chained_rosenbrock_closure_matches_oracle(cgo):
    """BASELINE config 1 in its chained form (test_funcs.jl:50-57) through the product's host engine: the objective as
    a closure (the numpy restatement) on the engine side, the C restatement on the oracle side."""
    from _cases import N
    n = 1000
    for beta, kw in (("PolakRibiere", dict(c2=0.1, max_iters=6)), ("HagerZhang", dict(ls="WolfeBisection", cond="Wolfe", c1=1e-3, c2=0.9, ls_max_iters=100, max_iters=10))):
        ref = run_oracle(Case("chain", "rosenbrock_chained", n, np.tile([-1.2, 1.0], n // 2), beta=beta, **kw))
        got = run_hostsim(Case("chain-closure", "closure", n, np.tile([-1.2, 1.0], n // 2), beta=beta, extra={"fdf": N.rosenbrock_chained}, **kw), points=1)
        assert_parity(got, ref, 1e-10, beta)


@pytest.mark.parametrize("c", parity_cases(sizes=(64,), small_only=True), ids=lambda c: c.name)
def test_engine_single_point_matches_multi_point(cgo, c):
    """Speculation must not change anything but the number of launches."""
    multi, single = run_hostsim(c), run_hostsim(c, chunk=-1)
    assert first_divergence(multi, single) is None
    assert np.array_equal(multi.minimizer, single.minimizer) and multi.objective == single.objective
    assert multi.total_fdf_evals == single.total_fdf_evals and multi.total_launches <= single.total_launches


@pytest.mark.parametrize("c", parity_cases(small_only=True), ids=lambda c: c.name)
def test_device_controller_changes_nothing(cgo, c):
    """The on-device controller (cgo_ctl.hpp: ctl_step, compiled for gfx950 in the product and run
    here by the test double) arms launches ahead of the host; the engine replays its records after a
    bit-for-bit check of every launch argument.  Trajectory, evaluation counts and results must be
    IDENTICAL to the host-driven run, for any depth and any iterate() slicing."""
    base = run_hostsim(c)
    for depth, chunk, pts in ((1, 0, 3), (4, 0, 3), (16, 0, 3), (5, 3, 3), (3, -1, 3), (4, 0, 7), (8, 0, 5)):
        st = {}
        if pts != 3:
            base = run_hostsim(c, points=pts)
        got = run_hostsim(c, chunk=chunk, ctl_depth=depth, ctl_stats=st, points=pts)
        assert first_divergence(got, base) is None, (depth, chunk)
        assert np.array_equal(got.minimizer, base.minimizer) and got.objective == base.objective
        assert got.status == base.status and got.iters_ran == base.iters_ran
        assert got.total_fdf_evals == base.total_fdf_evals
        assert np.array_equal(got.trace_objective_evals, base.trace_objective_evals)
        if chunk == 0:
            assert got.total_launches == base.total_launches, (depth, got.total_launches, base.total_launches)


def test_device_controller_serves_first_trial_streaks(cgo):
    """HagerZhang + weak Wolfe on a well-conditioned quadratic accepts most first trials: the
    controller must actually run ahead there (not merely be harmless)."""
    n = 512
    c = Case("ctl-streak", "quad_diag", n, np.ones(n), beta="HagerZhang", D=quad_D(n, 1.0, 20.0), eps=1e-12,
             max_iters=40, ls="WolfeBisection", cond="Wolfe", c1=1e-3, c2=0.9)
    st = {}
    got = run_hostsim(c, ctl_depth=8, ctl_stats=st)
    assert first_divergence(got, run_hostsim(c)) is None
    assert st["served"] >= got.iters_ran // 2, st


@pytest.mark.parametrize("want,c", status_cases(), ids=lambda v: v.name if isinstance(v, Case) else str(v))
def test_device_controller_status_paths(cgo, want, c):
    for pts in (3, 7):
        base, got = run_hostsim(c, points=pts), run_hostsim(c, ctl_depth=6, points=pts)
        assert got.status == base.status and got.iters_ran == base.iters_ran
        assert got.total_fdf_evals == base.total_fdf_evals
        assert np.array_equal(got.minimizer, base.minimizer, equal_nan=True)
    base, got = run_hostsim(c), run_hostsim(c, ctl_depth=6)
    assert got.status == base.status and got.iters_ran == base.iters_ran
    assert got.total_fdf_evals == base.total_fdf_evals
    assert np.array_equal(got.minimizer, base.minimizer, equal_nan=True)


@pytest.mark.parametrize("c", parity_cases(small_only=True) + backtracking_cases(), ids=lambda c: c.name)
def test_engine_five_point_speculation_changes_only_launch_counts(cgo, c):
    """5 / 7 trial steps per launch (requested, both candidates, then one / two more levels along the
    two paths that close in on the requested step)."""
    three = run_hostsim(c)
    for pts in (5, 7):
        more = run_hostsim(c, points=pts)
        assert first_divergence(more, three, 1e-12) is None
        assert np.array_equal(more.minimizer, three.minimizer) and more.objective == three.objective
        assert more.total_fdf_evals == three.total_fdf_evals and more.total_launches <= three.total_launches


@pytest.mark.parametrize("c", broyden_cases(), ids=lambda c: c.name)
def test_broyden_family_is_steepest_descent(cgo, c):
    """The reference's dense BroydenFamily update leaves B = I up to rounding (s = B\\y ⇒ Bs = y, v = 0;
    qn_flavours.jl:81-87).  Both oracles carry the dense n×n algebra as written; the engine's u = −g
    must reproduce their trajectories."""
    from _cases import run_numpy
    ref, alt, got = run_oracle(c), run_numpy(c), run_hostsim(c)
    rt = 1e-12 if c.ls == "Backtracking" else 0.0
    assert_parity(alt, ref, 1e-10, c.name, step_rtol=rt)
    assert_parity(got, ref, 1e-10, c.name, step_rtol=rt)
    with pytest.raises(AssertionError):
        cgo.setupBroydenFamily(-1.0, c.n)     # qn_flavours.jl:57


@pytest.mark.parametrize("c", backtracking_cases(), ids=lambda c: c.name)
def test_engine_backtracking_matches_oracle(cgo, c):
    """geometric.jl restated bug for bug; steps match to rounding (the first one is |ϕ₀|/u·u)."""
    assert_parity(run_hostsim(c), run_oracle(c), 1e-10, c.name, step_rtol=1e-12)


@pytest.mark.parametrize("want,c", status_cases(), ids=lambda v: v.name if isinstance(v, Case) else str(v))
def test_engine_status_paths(cgo, want, c):
    got, ref = run_hostsim(c), run_oracle(c)
    assert got.status == ref.status and got.iters_ran == ref.iters_ran
    if want is not None:
        assert got.status == want
    assert len(got.trace_objective) == got.iters_ran
    assert got.total_fdf_evals == ref.total_fdf_evals      # same number of objective evaluations
    if np.all(np.isfinite(ref.minimizer)):
        assert rel(got.minimizer, ref.minimizer) <= 1e-10   # last good iterate (optim.jl:93-104)


def test_engine_resumable_chunks_give_same_trajectory(cgo):
    """iterate(k) slices (what bench.py's warm-up/timed split uses) must not change the math."""
    n = 1000
    c = Case("chunks", "quad_diag", n, np.ones(n), beta="DaiYuan", D=quad_D(n), eps=1e-9, max_iters=25)
    whole = run_hostsim(c)
    for chunk in (1, 3, 7):
        part = run_hostsim(c, chunk=chunk)
        assert first_divergence(part, whole) is None
        assert np.array_equal(part.minimizer, whole.minimizer) and part.objective == whole.objective
        assert part.status == whole.status and part.iters_ran == whole.iters_ran


def test_engine_launch_count_is_one_per_accepted_first_trial(cgo):
    """Steady state = ONE fused launch per outer iteration when the first trial is accepted."""
    n = 64
    c = Case("launches", "quad_diag", n, np.ones(n), beta="DaiYuan", D=quad_D(n), eps=1e-12, max_iters=30)
    r = run_hostsim(c, chunk=-1)                       # single-point launches
    trials = int(r.trace_objective_evals.sum())
    # init + first trial + (per iteration: trials beyond the speculative one) + one accept launch each
    assert r.total_launches == 1 + trials + 1
    assert r.total_fdf_evals == 1 + trials
    m = run_hostsim(c)                                 # 3-point launches walk two tree levels each
    assert m.total_fdf_evals == 1 + trials and m.total_launches < r.total_launches
    # PR with c2 = 0.1 needs 1–4 trials per iteration: at most two launches each
    c = Case("launches-pr", "quad_diag", 1000, np.ones(1000), beta="PolakRibiere", D=quad_D(1000), eps=1e-12, max_iters=60, c2=0.1)
    m = run_hostsim(c)
    assert m.total_launches <= 2 + 2 * m.iters_ran + 12  # (+ the first, 10-trial line search from a = 1)


def test_engine_lbfgs_matches_oracle(cgo):
    n = 64
    c = Case("lbfgs", "rosenbrock_paired", n, rosen_x0(n), beta="LBFGS", m=10, max_iters=12, c2=0.5)
    assert_parity(run_hostsim(c), run_oracle(c), 1e-10, c.name)
    c = Case("lbfgs-q", "quad_diag", 1000, np.ones(1000), beta="LBFGS", m=4, D=quad_D(1000), eps=1e-9, max_iters=15, c2=0.9)
    assert_parity(run_hostsim(c), run_oracle(c), 1e-10, c.name)


def test_engine_lbfgs_gram_form_equals_two_loop(cgo):
    """Vector-free (Gram) L-BFGS direction vs the chained two-loop launches: same trajectory."""
    for c in (Case("lbfgs-r", "rosenbrock_paired", 64, rosen_x0(64), beta="LBFGS", m=10, max_iters=15, c2=0.5),
              Case("lbfgs-q", "quad_diag", 1000, np.ones(1000), beta="LBFGS", m=3, D=quad_D(1000), eps=1e-9, max_iters=20, c2=0.9),
              Case("lbfgs-bt", "rosenbrock_paired", 64, rosen_x0(64), beta="LBFGS", m=5, max_iters=15, ls="Backtracking",
                   c1=1e-3, discount=0.5, ls_max_iters=100)):
        gram, two = run_hostsim(c), run_hostsim(c, chunk=-1)
        assert first_divergence(gram, two, 1e-12) is None, c.name
        assert gram.status == two.status and gram.iters_ran == two.iters_ran
        assert rel(gram.minimizer, two.minimizer) <= 1e-10
        assert_parity(gram, run_oracle(c), 1e-10, c.name, step_rtol=1e-12)


def test_engine_lbfgs_one_ring_pass_protocol(cgo):
    """The product engine's one-ring-pass L-BFGS iteration (cgo_engine.cpp: lbfgs_push_spec → non-finite test →
    lbfgs_push_commit(direction_follows) → the direction pass that carries the state update; materialize() when a speculated
    trial's sums cannot be used) over the test double, which implements the backend's side of the protocol with plain loops:
    against the oracle, against the two-pass form, with the state update applied at once or left to the next direction pass,
    in one piece and in iterate() slices; the speculated pushes are exactly the iterations whose line search took its first
    trial, and a deferred update never has to be applied early except by the download at the end."""
    cases = (Case("lb1-r", "rosenbrock_paired", 64, rosen_x0(64), beta="LBFGS", m=10, max_iters=15, c2=0.5),
             Case("lb1-q", "quad_diag", 1001, np.ones(1001), beta="LBFGS", m=3, D=quad_D(1001), eps=1e-9, max_iters=25, c2=0.9),
             Case("lb1-q-tight", "quad_diag", 1000, np.ones(1000), beta="LBFGS", m=6, D=quad_D(1000), eps=1e-9, max_iters=25, c2=0.1),
             Case("lb1-r-wolfe", "rosenbrock_paired", 200, rosen_x0(200), beta="LBFGS", m=4, max_iters=14,
                  ls="WolfeBisection", c1=1e-3, c2=0.9, ls_max_iters=100),
             Case("lb1-bt", "rosenbrock_paired", 64, rosen_x0(64), beta="LBFGS", m=5, max_iters=15, ls="Backtracking",
                  c1=1e-3, discount=0.5, ls_max_iters=100))
    for c in cases:
        ref = run_oracle(c)
        two = run_hostsim(c)
        rt = 1e-12 if c.ls == "Backtracking" else 0.0
        for mode in (1, 2):
            for chunk in (0, 1, 3):
                st = {}
                one = run_hostsim(c, lbfgs_spec=mode, lbfgs_spec_stats=st, chunk=chunk)
                assert_parity(one, ref, 1e-10, f"{c.name} mode={mode} chunk={chunk}", step_rtol=rt)
                assert first_divergence(one, two, rt) is None and one.status == two.status and one.iters_ran == two.iters_ran
                assert rel(one.minimizer, two.minimizer) <= 1e-11 and rel(one.gradient, two.gradient) <= 1e-9
                if c.ls == "Backtracking":      # no first step to speculate on (its first step is itself a reduction): the two-pass form
                    assert st["pushes"] == 0
                    continue
                first_accepted = int(np.sum(np.asarray(one.trace_objective_evals)[1:] == 1))
                assert st["pushes"] == first_accepted >= 1, (c.name, st, list(one.trace_objective_evals))
                if mode == 2:   # every speculated push but possibly the solve's last one rode in the following direction pass
                    assert st["flushed"] == 0 and st["pushes"] - 1 <= st["rode"] <= st["pushes"], (c.name, st)
                    assert one.total_launches < two.total_launches - st["rode"], (c.name, one.total_launches, two.total_launches, st)
                else:
                    assert st["rode"] == 0 and st["flushed"] == 0


def test_engine_two_phase_objective_over_the_test_double(cgo):
    """The engine's branches for a two-phase objective (log-sum-exp: trials return ϕ, dϕ only; g⁺ of the accepted step is
    written by materialize(), or formed by the L-BFGS push itself with x and g untouched until lbfgs_push_commit — the
    non-finite test of optim.jl:107-121 sits in between —, or never stored at all when the direction pass had speculated on
    the accepted step) over the test double's plain loops, against the oracle: CG flavours, and L-BFGS in every combination
    of fused / plain push and two-pass / one-pass iteration, in one piece and in slices."""
    from _cases import O
    def x0(n, scale=5.0):
        return scale * O.fill_uniform(n, 24, -1.0, 1.0)
    for beta in ("DaiYuan", "HagerZhang", "PolakRibiere"):
        c = Case(f"sim-lse-{beta}", "lse", 257, x0(257), beta=beta, lam=1e-4, max_iters=10, c2=0.1 if beta == "PolakRibiere" else 0.8, eps=1e-12)
        assert_parity(run_hostsim(c), run_oracle(c), 1e-10, c.name)
    cases = (Case("sim-lse-lbfgs10", "lse", 1000, x0(1000), beta="LBFGS", m=10, lam=1e-5, max_iters=14, c2=0.9, eps=1e-12),
             Case("sim-lse-lbfgs3-tight", "lse", 333, x0(333), beta="LBFGS", m=3, lam=1e-3, max_iters=12, c2=0.1, eps=1e-12),
             Case("sim-lse-lbfgs4-wolfe", "lse", 64, x0(64, 30.0), beta="LBFGS", m=4, lam=1e-2, max_iters=10, eps=1e-12,
                  ls="WolfeBisection", c1=1e-3, c2=0.9, ls_max_iters=100))
    for c in cases:
        ref = run_oracle(c)
        base = run_hostsim(c, fuse_grad=False)                 # materialize() + plain push + direction: the round-2 flow
        assert_parity(base, ref, 1e-10, c.name + " (plain)")
        for fuse in (True, False):
            for mode in (0, 1, 2):
                for chunk in (0, 2):
                    st = {}
                    got = run_hostsim(c, fuse_grad=fuse, lbfgs_spec=mode, lbfgs_spec_stats=st, chunk=chunk)
                    what = f"{c.name} fuse={fuse} spec={mode} chunk={chunk}"
                    assert_parity(got, ref, 1e-10, what)
                    assert first_divergence(got, base) is None and got.status == base.status and got.iters_ran == base.iters_ran, what
                    assert rel(got.minimizer, base.minimizer) <= 1e-11 and rel(got.gradient, base.gradient) <= 1e-9, what
                    first_accepted = int(np.sum(np.asarray(got.trace_objective_evals)[1:] == 1))
                    assert st["pushes"] == (first_accepted if mode else 0), (what, st)
                    assert st["fused"] == ((got.iters_ran - st["pushes"]) if fuse else 0), (what, st)


def test_engine_returns_the_last_good_iterate_when_a_pushed_gradient_is_not_finite(cgo):
    """optim.jl:107-121: a proposed iterate whose gradient norm is not finite ends the solve with the LAST GOOD iterate, its
    gradient and its objective.  With the push forming g⁺ itself the state must not have moved when the engine gets to that
    test — x advances out of place, x / g change only in lbfgs_push_commit.  The test double poisons g⁺ of its k-th fused
    push: the solve must end :non_finite_objective_or_gradient_proposed after k − 1 iterations with exactly the iterate, the
    gradient and the trace of the same solve stopped there by max_iters."""
    import ctypes as C
    from _cases import O
    L = sim_lib()
    L.sim_set_poison_push.restype = None
    L.sim_set_poison_push.argtypes = [C.c_int]
    n = 300
    x0 = 5.0 * O.fill_uniform(n, 24, -1.0, 1.0)
    base = dict(beta="LBFGS", m=5, lam=1e-4, c2=0.9, eps=1e-12)
    for k in (1, 2, 4):
        L.sim_set_poison_push(k)
        try:
            bad = run_hostsim(Case("poison", "lse", n, x0, max_iters=12, **base))
        finally:
            L.sim_set_poison_push(0)
        assert bad.status == "non_finite_objective_or_gradient_proposed" and bad.iters_ran == k - 1, (k, bad.status, bad.iters_ran)
        if k == 1:
            f0, g0 = O.objective("lse", lam=1e-4)(x0)
            assert np.array_equal(bad.minimizer, x0) and rel(bad.gradient, g0) <= 1e-14 and abs(bad.objective - f0) <= 1e-13 * abs(f0)
            assert len(bad.trace_objective) == 0
        else:
            good = run_hostsim(Case("stop", "lse", n, x0, max_iters=k - 1, **base))
            assert good.status == "max_iters_reached" and good.iters_ran == k - 1
            assert np.array_equal(bad.minimizer, good.minimizer) and np.array_equal(bad.gradient, good.gradient)
            assert bad.objective == good.objective and np.array_equal(bad.trace_objective, good.trace_objective)
            assert np.array_equal(bad.trace_step_size, good.trace_step_size)


def test_beta_from_scalars_kat(cgo):
    """The engine's scalar β formulas against the hand-derived values."""
    import ctypes as C
    from cgo_amd import _lib
    L = sim_lib()
    t = np.array([-1.0, 5.0, 1.0, 13.0, 9.0, 4.0])  # g⁺·u, g⁺·g⁺, g⁺·g, y·y, u·y, y·g⁺
    want = {0: 62 / 81, 1: 62 / 81, 2: 4 / 9, 3: -4 / 9, 4: 2 / 5, 5: 4 / 9, 6: 5 / 9}
    for kind, w in want.items():
        b = _lib.BetaConfig(kind, 0, 0.1)
        got = L.sim_beta_from_scalars(C.byref(b), t.ctypes.data_as(_lib.dp), -10.0, 10.0, 10.0)
        assert abs(got - w) < 4e-16, (kind, got, w)


WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
import torch
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from _cases import Case, quad_D, run_hostsim, run_oracle, rel, first_divergence
from _suite import rosen_x0
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{PORT}", rank=RANK, world_size=WORLD)
def allgather(send):
    t = torch.from_numpy(send.copy())
    out = [torch.empty_like(t) for _ in range(WORLD)]
    dist.all_gather(out, t)
    return torch.cat(out).numpy()
n = 1001
cases = [Case("q-DY", "quad_diag", n, np.ones(n), beta="DaiYuan", D=quad_D(n), eps=1e-9, max_iters=16),
         Case("q-HZ-W", "quad_diag", n, np.ones(n), beta="HagerZhang", D=quad_D(n), eps=1e-9, max_iters=16,
              ls="WolfeBisection", c1=1e-3, c2=0.9, ls_max_iters=100),
         Case("r-SA", "rosenbrock_paired", 1000, rosen_x0(1000), beta="SallehAlhawarat", max_iters=12)]
import cgo_amd as cgo
for c in cases:
    got = run_hostsim(c, RANK, WORLD, allgather)
    ref = run_oracle(c)
    off, nloc = cgo.shard_extent(c.n, RANK, WORLD)
    assert first_divergence(got, ref) is None, c.name
    assert got.status == ref.status and got.iters_ran == ref.iters_ran
    assert rel(got.minimizer, ref.minimizer[off:off+nloc]) <= 1e-10, c.name
    assert abs(got.objective - ref.objective) <= 1e-10 * abs(ref.objective), c.name
    # replicated control flow: every rank must hold bitwise identical scalars
    t = torch.tensor([got.objective] + list(got.trace_step_size)); ts = [torch.empty_like(t) for _ in range(WORLD)]
    dist.all_gather(ts, t)
    assert all(torch.equal(ts[0], q) for q in ts), "ranks disagree on reduced scalars"
dist.destroy_process_group()
print("RANK", RANK, "OK")
'''


def test_sharded_engine_world2_gloo(cgo, tmp_path):
    """N>1 path on CPU: 2 ranks (gloo), contiguous shards, rank-ordered scalar all-gather
    through the same callback ABI (cgo_allgather_fn) the product exposes."""
    sim_lib()
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for rank in range(2):
        code = f"ROOT={ROOT!r}; PORT={port}; RANK={rank}; WORLD=2\n" + WORKER
        p = tmp_path / f"w{rank}.py"
        p.write_text(code)
        procs.append(subprocess.Popen([sys.executable, str(p)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK {rank} OK" in o, o[-3000:]


def test_sharded_engine_world8_gloo(cgo, tmp_path):
    """The 8-rank layout of BASELINE config 5 on CPU: 8 processes (gloo), contiguous even-aligned shards (the odd tail
    element on the last rank), the rank-ordered merge of eight scalar blocks through the callback ABI — every rank
    must take the oracle's decisions and hold the same scalars bit for bit."""
    sim_lib()
    port = 31500 + (os.getpid() % 2000)
    procs = []
    for rank in range(8):
        code = f"ROOT={ROOT!r}; PORT={port}; RANK={rank}; WORLD=8\n" + WORKER
        p = tmp_path / f"w8_{rank}.py"
        p.write_text(code)
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(p)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK {rank} OK" in o, o[-3000:]


# ---------------------------------------------------------------- resident solver loop (csrc/cgo_resident.hpp)
def _same_solve(got, base):
    assert first_divergence(got, base) is None
    assert np.array_equal(got.log_phi, base.log_phi, equal_nan=True) and np.array_equal(got.log_dphi, base.log_dphi, equal_nan=True)
    assert np.array_equal(got.minimizer, base.minimizer, equal_nan=True) and (got.objective == base.objective or (np.isnan(got.objective) and np.isnan(base.objective)))
    assert np.array_equal(got.gradient, base.gradient, equal_nan=True)
    assert got.status == base.status and got.iters_ran == base.iters_ran
    assert got.total_fdf_evals == base.total_fdf_evals
    for a, b in ((got.trace_objective, base.trace_objective), (got.trace_grad_norm, base.trace_grad_norm),
                 (got.trace_step_size, base.trace_step_size), (got.trace_objective_evals, base.trace_objective_evals)):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("c", parity_cases(small_only=True), ids=lambda c: c.name)
def test_resident_loop_is_the_host_loop(cgo, c):
    """`res_iterate` — the loop every thread of the resident kernel runs (outer loop, both bisection line searches, getβ,
    the fused accept + direction + first trials) — driven over the test double must ask for EXACTLY the launches the
    host engine asks for and take exactly its decisions: identical trial log, trace, results and launch count, for
    1 / 3 / 7 trial steps per pass and any slicing; and it must meet the oracle."""
    for pts in (3, 1, 7):
        base = run_hostsim(c, points=pts)
        st = {}
        got = run_hostsim(c, points=pts, resident=True, resident_stats=st)
        _same_solve(got, base)
        assert got.total_launches == base.total_launches, (pts, got.total_launches, base.total_launches)
        assert st["iters"] >= min(base.iters_ran, 1), st     # the slices did the work, not the fallback
        for chunk in (1, 3):
            _same_solve(run_hostsim(c, points=pts, resident=True, chunk=chunk), base)
    assert_parity(run_hostsim(c, resident=True), run_oracle(c), 1e-10, c.name)


@pytest.mark.parametrize("want,c", status_cases(), ids=lambda v: v.name if isinstance(v, Case) else str(v))
def test_resident_loop_hands_every_other_outcome_to_the_host(cgo, want, c):
    """Every status but :success / :max_iters_reached is the host's: the resident loop stops BEFORE touching x or u and the
    host engine runs that iteration — same status, same iteration count, same last-good iterate as without it."""
    if c.ls == "Backtracking":
        pytest.skip("Backtracking is host-driven")
    base = run_hostsim(c)
    got = run_hostsim(c, resident=True)
    assert got.status == base.status and (want is None or got.status == want)
    _same_solve(got, base)


@pytest.mark.parametrize("c", reset_cases(), ids=lambda c: c.name)
def test_resident_loop_through_the_wolfe_reset(cgo, c):
    """The bracket collapse of wolfe.jl:122-130 needs vector work (‖u + g‖, u ← −g): handed back mid-solve, then the
    slices resume — 200 iterations, bitwise the host-driven solve."""
    st = {}
    got = run_hostsim(c, resident=True, resident_stats=st)
    _same_solve(got, run_hostsim(c))
    assert st["host"] >= 1 and st["iters"] > 100, st


def test_resident_loop_drains_a_small_trial_log(cgo):
    """A slice stops when its trial log runs low (RES_LOG_FULL) and the next one carries on; an iteration whose line
    search does not fit at all is run by the host."""
    n = 64
    c = Case("res-log", "quad_diag", n, np.ones(n), beta="PolakRibiere", D=quad_D(n), eps=1e-12, max_iters=40, c2=0.1)
    base = run_hostsim(c)
    for cap in (2049, 2060, 4096):    # RES_LOG_MARGIN = 2048: one, a dozen, many free entries per slice
        st = {}
        _same_solve(run_hostsim(c, resident=True, resident_log_cap=cap, resident_stats=st), base)
        assert st["slices"] >= 2 or cap == 4096, (cap, st)
