"""GPU tier: the resident solver (csrc/cgo_resident.hpp + cgo_kernels_resident.hip.hpp) — a slice of whole outer
iterations in ONE launch, x / u / D in LDS, every workgroup running the reference's line search itself.

Held to the same bar as every other path: identical step log, status, iteration count, trials per iteration against
oracle/cgo_oracle.c, then ≤ 1e-10 on iterate and objective; bit-reproducible run to run and for any slicing; the
host-driven launches (CGO_RESIDENT=0) remain the reference implementation of every iteration it hands back.
The scalar loop itself (res_iterate) is held bitwise to the host engine on the CPU tier (tests/test_hostsim.py).
"""
import numpy as np
import pytest

from _cases import Case, assert_parity, first_divergence, quad_D, rel, relf, run_gpu, run_oracle, _product_structs, gpu_objective, Out
from _suite import parity_cases, reset_cases, status_cases, rosen_x0

pytestmark = pytest.mark.gpu


def run_resident(c, ctx, chunk=0):
    """run_gpu + how much of the solve ran inside resident slices."""
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
        stats = s.resident_stats()
    finally:
        s.close()
        obj.close()
    return Out(r.objective, r.minimizer, r.gradient, r.iters_ran, r.status, r.trace.objective, r.trace.grad_norm, r.trace.step_size,
               r.trace.objective_evals, la, lp, ld, r.total_fdf_evals, r.total_launches), stats


def eligible(c):
    return (c.objective in ("quad_diag", "rosenbrock_paired", "booth") and c.beta != "LBFGS"
            and c.ls in ("StrongWolfeBisection", "WolfeBisection"))


def same_bits(a, b):
    assert first_divergence(a, b) is None
    assert np.array_equal(a.log_phi, b.log_phi, equal_nan=True) and np.array_equal(a.log_dphi, b.log_dphi, equal_nan=True)
    assert np.array_equal(a.minimizer, b.minimizer, equal_nan=True) and np.array_equal(a.gradient, b.gradient, equal_nan=True)
    assert a.status == b.status and a.iters_ran == b.iters_ran and a.total_fdf_evals == b.total_fdf_evals
    assert np.array_equal(a.trace_objective, b.trace_objective, equal_nan=True)
    assert np.array_equal(a.trace_grad_norm, b.trace_grad_norm, equal_nan=True)
    assert np.array_equal(a.trace_objective_evals, b.trace_objective_evals)


@pytest.mark.parametrize("c", parity_cases(), ids=lambda c: c.name)
def test_resident_trajectory_parity_vs_oracle(cgo, gpu_ctx, c, monkeypatch):
    """Every parity case under the default policy: the eligible ones must have run their iterations inside slices."""
    monkeypatch.delenv("CGO_RESIDENT", raising=False)
    got, (slices, iters) = run_resident(c, gpu_ctx)
    assert_parity(got, run_oracle(c), 1e-10, c.name)
    if eligible(c):
        assert slices >= 1 and iters >= got.iters_ran - 2, (slices, iters, got.iters_ran)
        assert got.total_launches <= slices + 6 + 4 * (got.iters_ran - iters), (got.total_launches, slices)


@pytest.mark.parametrize("pts", [1, 3, 7])
@pytest.mark.parametrize("c", parity_cases(sizes=(1000, 100003)), ids=lambda c: c.name)
def test_resident_points_per_pass(cgo, gpu_ctx, c, pts, monkeypatch):
    """1, 3 or 7 trial steps per pass: speculation changes passes, never a step."""
    monkeypatch.setenv("CGO_RES_POINTS", str(pts))
    got, _ = run_resident(c, gpu_ctx)
    assert_parity(got, run_oracle(c), 1e-10, f"{c.name} pts={pts}")


@pytest.mark.parametrize("chunk_elems", [2, 8, 64, 1024])
def test_resident_many_workgroups_exchange(cgo, gpu_ctx, chunk_elems, monkeypatch):
    """The all-gather between workgroups under stress: tiny chunks put 16 … 256 workgroups on problems of a few hundred
    to a few thousand elements (thousands of exchange rounds, four rotating row buffers), odd sizes put the tail element
    in the last workgroup.  Against the oracle; and the same solve twice gives the same bits."""
    monkeypatch.setenv("CGO_RES_CHUNK", str(chunk_elems))
    for n in (min(256 * chunk_elems, 4001), min(256 * chunk_elems - 1, 3001), 31 * chunk_elems + 1):
        D = quad_D(n)
        for c in (Case(f"xq{n}-PR", "quad_diag", n, np.ones(n), beta="PolakRibiere", D=D, eps=1e-9, max_iters=16, c2=0.1),
                  Case(f"xq{n}-HZ-W", "quad_diag", n, np.ones(n), beta="HagerZhang", D=D, eps=1e-9, max_iters=16, ls="WolfeBisection",
                       cond="Wolfe", c1=1e-3, c2=0.9, ls_max_iters=100)):
            got, (slices, iters) = run_resident(c, gpu_ctx)
            assert_parity(got, run_oracle(c), 1e-10, c.name)
            assert iters >= got.iters_ran - 2
    # reproducible: the same solve twice, bit for bit
    n = min(256 * chunk_elems, 4001)
    c = Case("xq-rep", "quad_diag", n, np.ones(n), beta="DaiYuan", D=quad_D(n), eps=1e-9, max_iters=30, c2=0.8)
    a, _ = run_resident(c, gpu_ctx)
    b, _ = run_resident(c, gpu_ctx)
    same_bits(a, b)


@pytest.mark.parametrize("n", [4096, 4098, 100003, 1000000, 1500001])
def test_resident_sizes_up_to_the_lds_of_the_chip(cgo, gpu_ctx, n, monkeypatch):
    """One workgroup (n ≤ 4096) … 245 workgroups of 4082 (n = 1e6, BASELINE config 2's size) … 256 workgroups with
    5860 elements each (n = 1.5e6: the chunk grows to what the LDS holds)."""
    D = quad_D(n)
    for c in (Case(f"rq{n}-PR", "quad_diag", n, np.ones(n), beta="PolakRibiere", D=D, eps=1e-200, max_iters=10, c2=0.1),
              Case(f"rr{n}-HZ", "rosenbrock_paired", n - (n & 1), rosen_x0(n - (n & 1)), beta="HagerZhang", max_iters=8, ls="WolfeBisection",
                   cond="Wolfe", c1=1e-3, c2=0.9, ls_max_iters=100)):
        got, (slices, iters) = run_resident(c, gpu_ctx)
        assert_parity(got, run_oracle(c), 1e-10, c.name)
        assert iters == got.iters_ran and slices <= 3, (slices, iters)


def test_resident_slicing_and_reruns_are_invisible(cgo, gpu_ctx, monkeypatch):
    """iterate(k) slices, a second solve on the same solver's buffers: bitwise the same solve."""
    n = 20000
    c = Case("rs", "quad_diag", n, np.ones(n), beta="PolakRibiere", D=quad_D(n), eps=1e-200, max_iters=60, c2=0.1)
    base, (s0, i0) = run_resident(c, gpu_ctx)
    assert i0 == 60
    for chunk in (1, 7, 25):
        got, (s1, i1) = run_resident(c, gpu_ctx, chunk=chunk)
        same_bits(got, base)
        assert i1 == 60 and s1 >= 60 // chunk


def test_resident_against_the_host_driven_launches(cgo, gpu_ctx, monkeypatch):
    """Same steps, same decisions as the launch-per-trial path (different summation order: not the same bits), in a
    fraction of the launches."""
    n = 1000
    c = Case("rh", "rosenbrock_paired", n, np.tile([-1.2, 1.0], n // 2), beta="PolakRibiere", max_iters=5, c2=0.1)
    res, (slices, iters) = run_resident(c, gpu_ctx)
    monkeypatch.setenv("CGO_RESIDENT", "0")
    host, (s2, i2) = run_resident(c, gpu_ctx)
    assert s2 == 0 and i2 == 0 and slices >= 1 and iters == 5
    assert first_divergence(res, host) is None and res.status == host.status
    assert rel(res.minimizer, host.minimizer) <= 1e-10 and relf(res.objective, host.objective) <= 1e-10
    assert res.total_launches < host.total_launches / 3, (res.total_launches, host.total_launches)


@pytest.mark.parametrize("want,c", status_cases(), ids=lambda v: v.name if isinstance(v, Case) else str(v))
def test_resident_status_paths(cgo, gpu_ctx, want, c, monkeypatch):
    """Every outcome other than :success is the host's: same status, same last-good iterate as the oracle."""
    got, _ = run_resident(c, gpu_ctx)
    ref = run_oracle(c)
    assert got.status == ref.status and (want is None or got.status == want), (got.status, ref.status)
    assert got.iters_ran == ref.iters_ran
    if np.all(np.isfinite(ref.minimizer)):
        assert rel(got.minimizer, ref.minimizer) <= 1e-9


@pytest.mark.parametrize("c", reset_cases(), ids=lambda c: c.name)
def test_resident_through_the_wolfe_reset(cgo, gpu_ctx, c, monkeypatch):
    """wolfe.jl:122-130 needs vector work: handed back mid-solve, slices resume afterwards.  n = 2: every sum is ONE pair, so
    the resident passes and the host-driven launches (same fused bodies, same FMA sums) agree to the last bit over all 200
    iterations — through the collapse, the restart with the stale dϕ₀ and the getβ after it.  (Against the oracle the GPU's
    FMA sums part ways after ≈ 20 iterations of this chaotic horizon: test_wolfe_reset_long_horizon; the quirk itself is
    pinned by hand in tests/test_kat_quirks.py, KAT A.)"""
    got, (slices, iters) = run_resident(c, gpu_ctx)
    monkeypatch.setenv("CGO_RESIDENT", "0")
    host, _ = run_resident(c, gpu_ctx)
    same_bits(got, host)
    assert iters > 100 and slices >= 1   # (whether THIS trajectory meets the collapse depends on its last bits: SA does on the GPU, DY does not)


@pytest.mark.parametrize("wg", [0, 1, 4])
def test_resident_single_workgroup_give_up_never_leaves_a_mixed_state(cgo, gpu_ctx, wg, monkeypatch):
    """ONE workgroup gives up in the slice's last pass while its peers complete the slice (injected:
    CGO_RES_INJECT_GIVEUP, first slice only).  Before round 4 the peers wrote their chunks of x, u back, the culprit did not,
    and workgroup 0 could report a good slice over the mixture.  Now the slice's x, u go to other buffers and are swapped in
    only on the verdict of ALL workgroups: the slice is discarded whole, the launch-per-trial engine redoes it from the
    slice-start state, and the solve is bit for bit the host-driven one."""
    n = 20000                                                    # five workgroups of 4096
    c = Case("giveup", "quad_diag", n, np.ones(n), beta="PolakRibiere", D=quad_D(n), eps=1e-200, max_iters=40, c2=0.1)
    monkeypatch.setenv("CGO_RESIDENT", "0")
    host, _ = run_resident(c, gpu_ctx)
    monkeypatch.setenv("CGO_RESIDENT", "1")
    good, (s_good, i_good) = run_resident(c, gpu_ctx)
    assert i_good == 40
    monkeypatch.setenv("CGO_RES_INJECT_GIVEUP", str(wg))
    cgo_, _lib, cfg, ls = _product_structs(c)
    obj = gpu_objective(c, gpu_ctx)
    s = cgo_.Solver(obj, cfg, ls)
    try:
        s.enable_trial_log()
        s.set_x0(c.x0)
        s.start()
        while not s.iterate(1 << 40):
            pass
        r = s.results()
        la, lp, ld = s.trial_log()
        slices, iters = s.resident_stats()
        gave_up = s.resident_gave_up
    finally:
        s.close(); obj.close()
    got = Out(r.objective, r.minimizer, r.gradient, r.iters_ran, r.status, r.trace.objective, r.trace.grad_norm, r.trace.step_size,
              r.trace.objective_evals, la, lp, ld, r.total_fdf_evals, r.total_launches)
    assert gave_up == 1 and iters == 0, (gave_up, slices, iters)     # the slice was handed back whole; the solver left the resident path
    same_bits(got, host)


def test_resident_solves_survive_sharing_the_gpu(cgo, gpu_ctx):
    """Three processes on this one GPU run 245-workgroup resident solves at the same time (scripts/soak_resident.py).  Persistent
    launches of different processes can starve each other of CUs; a slice that cannot complete its exchange is given up
    unchanged, redone by the launch-per-trial engine, and the solver leaves the resident path — every solve ends with the
    undisturbed results either way."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_resident.py"), "3", "12", "1000000"],
                       capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ------------------------------------------------------------------ user-compiled (hiprtc) objectives in resident form
QUAD_BODY = "gi = p*x; fi = 0.5*(gi*x);"
ROSEN_STRUCT = """
struct UserObjective {
    static constexpr bool kParam = false;
    static constexpr bool kPairOnly = true;
    __device__ static inline void eval2(d2 x, d2, double, double &f, d2 &g) {
        const double t1 = x.y - x.x * x.x;
        const double t2 = 1.0 - x.x;
        f += 100.0 * (t1 * t1) + t2 * t2;
        g.x = -400.0 * (x.x * t1) - 2.0 * t2;
        g.y = 200.0 * t1;
    }
    __device__ static inline void eval1(double, double, double, double &, double &g) { g = 0.0; }
};
"""
QUARTIC_BODY = "const double x2 = x*x; fi = 0.25*(x2*x2) + 0.5*(p*x2) - s0*x; gi = x2*x + p*x - s0;"


def _solve(cgo, obj, x0, beta, ls, max_iters, eps=1e-12):
    cfg = cgo.setupCGConfig(eps, beta, cgo.EnableTrace(), max_iters=max_iters)
    s = cgo.Solver(obj, cfg, ls)
    s.enable_trial_log()
    s.set_x0(x0)
    s.start()
    while not s.iterate(1 << 40):
        pass
    r, log, st = s.results(), s.trial_log(), s.resident_stats()
    s.close()
    return r, log, st


def test_resident_user_compiled_objective_matches_builtin_bit_for_bit(cgo, gpu_ctx):
    """The user's element-wise source compiled at run time carries its own k_resident<UserObjective, 3> (cgo_rtc.hip): the same
    expression must reproduce the ahead-of-time resident kernel exactly — quadratic through the body form (one and several
    workgroups, odd size), Rosenbrock through the functor form."""
    ls = cgo.setupStrongWolfeBisection(1e-5, 0.1)
    for n in (4097, 20001):
        D, x0 = quad_D(n), np.ones(n)
        for beta in (cgo.PolakRibiere(), cgo.HagerZhang()):
            a, la, sa = _solve(cgo, cgo.QuadDiag(D), x0, beta, ls, 14)
            b, lb, sb = _solve(cgo, cgo.ElementwiseObjective(n, QUAD_BODY, param=D), x0, beta, ls, 14)
            assert sa[1] >= 12 and sb[1] >= 12, (sa, sb)                    # both ran their iterations inside slices
            assert np.array_equal(la[0], lb[0]) and np.array_equal(la[1], lb[1]) and a.status == b.status and a.iters_ran == b.iters_ran
            assert np.array_equal(a.minimizer, b.minimizer) and a.objective == b.objective and np.array_equal(a.gradient, b.gradient)
    n = 1000
    x0 = rosen_x0(n)
    lw = cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50)
    a, la, sa = _solve(cgo, cgo.RosenbrockPaired(n), x0, cgo.HagerZhang(), lw, 12)
    b, lb, sb = _solve(cgo, cgo.ElementwiseObjective(n, ROSEN_STRUCT), x0, cgo.HagerZhang(), lw, 12)
    assert sb[1] >= 10 and np.array_equal(la[0], lb[0]) and np.array_equal(a.minimizer, b.minimizer) and a.objective == b.objective


def test_resident_user_compiled_new_function_vs_oracle_closure(cgo, gpu_ctx):
    """A function no built-in covers (a double-well quartic with a parameter vector and a scalar), as device source on the
    resident path and as a closure in the oracle: same steps, ≤ 1e-10."""
    from oracle import oracle as O
    n = 3001
    D = O.fill_uniform(n, 3, -1.0, 2.0)
    s0 = 0.3

    def fdf(g, x):
        x2 = x * x
        g[:] = x2 * x + D * x - s0
        return float(np.sum(0.25 * (x2 * x2) + 0.5 * (D * x2) - s0 * x))
    x0 = 0.5 + O.fill_uniform(n, 5, -0.2, 0.2)
    obj = cgo.ElementwiseObjective(n, QUARTIC_BODY, param=D)
    obj.set_scalar(s0)
    got, log, st = _solve(cgo, obj, x0, cgo.HagerZhang(), cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50), 14)
    ref = O.minimizeobjective(O.python_objective(fdf), x0, O.cg_config(1e-12, O.beta_config("HagerZhang"), 14),
                              O.wolfe_bisection("Wolfe", 1e-3, 0.9, 0.0, 100, 1e12, 50), log_cap=10000)
    assert st[1] >= 12, st
    assert np.array_equal(log[0], ref.log_a) and got.status == ref.status and got.iters_ran == ref.iters_ran
    assert rel(got.minimizer, ref.minimizer) <= 1e-10 and relf(got.objective, ref.objective) <= 1e-10


# ------------------------------------------------------------------ the stencil objective in resident form (one workgroup)
def chain_x0(n, jitter=0.01, seed=7):
    from oracle import oracle as O
    return np.resize(np.tile([-1.2, 1.0], n // 2 + 1), n) + jitter * O.fill_uniform(n, seed, -1.0, 1.0)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 510, 511, 512, 513, 514, 1000, 1001, 4000, 4001])
def test_resident_chained_rosenbrock(cgo, gpu_ctx, n, monkeypatch):
    """BASELINE config 1 in its chained form (test_funcs.jl:50-57): the 3-point stencil objective resident in ONE workgroup —
    x and u in two LDS copies each, a pass reads one and writes the other.  Even and odd lengths, around the workgroup
    size, against the oracle; bit for bit the launch-per-trial stencil kernels (same window arithmetic, one workgroup
    there as well up to n = 1024)."""
    cases = [Case(f"rchain{n}-HZ-Wolfe", "rosenbrock_chained", n, chain_x0(n), beta="HagerZhang", max_iters=10, ls="WolfeBisection", cond="Wolfe",
                  c1=1e-3, c2=0.9, ls_max_iters=100),
             Case(f"rchain{n}-DY-SW", "rosenbrock_chained", n, chain_x0(n), beta="DaiYuan", max_iters=10, c2=0.8)]
    if n == 1000:
        cases.append(Case("rchain1000-PR-SW-config1", "rosenbrock_chained", 1000, np.tile([-1.2, 1.0], 500), beta="PolakRibiere", max_iters=6, c2=0.1))
    for c in cases:
        got, (slices, iters) = run_resident(c, gpu_ctx)
        ref = run_oracle(c)
        assert_parity(got, ref, 1e-10, c.name)
        assert rel(got.gradient, ref.gradient) <= 1e-9
        assert slices >= 1 and iters >= got.iters_ran - 2, (slices, iters, got.iters_ran)
        if n <= 1024:   # one workgroup on both paths: the same partition of the sums
            monkeypatch.setenv("CGO_RESIDENT", "0")
            host, _ = run_resident(c, gpu_ctx)
            monkeypatch.delenv("CGO_RESIDENT")
            same_bits(got, host)


def test_resident_chained_rosenbrock_too_large_keeps_its_launches(cgo, gpu_ctx):
    n = 100002
    c = Case("rchain-big", "rosenbrock_chained", n, chain_x0(n), beta="HagerZhang", max_iters=8, ls="WolfeBisection", cond="Wolfe", c1=1e-3, c2=0.9, ls_max_iters=100)
    got, (slices, iters) = run_resident(c, gpu_ctx)
    assert slices == 0 and iters == 0
    assert_parity(got, run_oracle(c), 1e-10, c.name)
