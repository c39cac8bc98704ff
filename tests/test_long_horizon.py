"""Long-horizon parity against the ARBITER build of the oracle (tests/_long_horizon.py).

`assert_parity` holds 1e-10 "for the same step sequence" over 5–12 iterations; bench.py times 100–250.  Here the product
(CPU tier: its host engine over the test double; GPU tier: libcgo_hip.so at the BASELINE sizes) and the double-precision
oracle(s) are walked along the trajectory the reference's algorithm defines when its reductions are accumulated in twice
the working precision and rounded once (oracle/cgo_oracle.c -DORC_EXACT_SUMS), over the horizon bench.py times:

  1. every implementation takes the arbiter's line-search branches for as long as the arbiter's own decision margin
     exceeds max(1e-9, 100 × the drift measured so far);
  2. the error-vs-exact curves (f, ‖g‖ per iteration; the iterate at checkpoints) are measured for all of them;
  3. the product's iterate error at every checkpoint is ≤ K × the double oracles' (other valid summation orders) + 1e-13,
     K = 10 (Polak–Ribière, L-BFGS) or 100 (Hager–Zhang).  Why not tighter: on Rosenbrock Hager–Zhang multiplies whatever the
     first iterations' roundings inject by ≈ 10³ per ten iterations, so the curves of ALL implementations run parallel on a log
     scale and what separates them is a constant factor fixed by iteration ≈ 8 — measured here between the oracle's own two
     summation orders (C loops vs OpenMP chunks): 2× at n = 1e5, 190× at n = 2e5.  The product evaluates getβ from the
     reduced sums ((y·g⁺ − m·u·g⁺)/R) where the reference rounds Σ(y − m·u)_i·(g⁺/R)_i element by element; that
     costs it a factor 3–10 at iteration 4, which then rides along (measured: 4–12× the larger of the two oracles).

The arbiter itself is pinned first: its dot product equals a __float128 accumulation of the same terms, rounded once, and
does not depend on the thread count.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import _long_horizon as LH
from _big_oracle import baseline_case
from _cases import run_hostsim

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
K_ITERATE = {"c2": 10.0, "c4": 10.0, "c5": 10.0, "c3": 100.0}     # part 3: factor on the double oracles' error (see above)
FLOOR = 1e-13


def _cores():
    sys.path.insert(0, ROOT)
    import bench
    return bench.usable_cores()


def oracle_child(config, n, iters, build, ckpts, tmp_path, threads):
    out = str(tmp_path / f"lh_{config}_{build}.npz")
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    r = subprocess.run([sys.executable, os.path.join(HERE, "_long_horizon.py"), config, str(n), str(iters), out, build,
                        ",".join(str(k) for k in ckpts)], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-800:]
    return LH.load(out)


def check(prod: LH.Traj, doubles, arb: LH.Traj, label: str, K: float):
    """The three-part statement → (JSON-able record, list of violations)."""
    rec = {"arbiter": dict(status=arb.status, iters_ran=arb.iters_ran, evaluations=len(arb.log_a),
                           min_margin=float(arb.log_margin.min()) if len(arb.log_margin) else None)}
    bad, cmps = [], {}
    for t in [prod] + list(doubles):
        c = LH.compare(t, arb)
        cmps[t.name] = c
        rec[t.name] = LH.summary(c)
        # part 1: where a trajectory leaves the arbiter's, the arbiter's margin at that decision is small against the drift so far
        if c["parted"]:
            drift = LH.drift_before(c)
            if c["arbiter_margin_where_parted"] is None or not c["arbiter_margin_where_parted"] <= max(1e-9, 100.0 * drift):
                bad.append(f"{label}/{t.name}: left the arbiter's step sequence at evaluation {c['evaluations_in_common']} "
                           f"(iteration {c['iterations_in_common']}) where the arbiter's margin is {c['arbiter_margin_where_parted']}, "
                           f"drift so far {drift:.3e}")
    # part 3: the product is as close to the arbiter as the double oracles are
    p = cmps[prod.name]
    for k, e in p["x_err"].items():
        refs = [cmps[t.name]["x_err"][k] for t in doubles if k in cmps[t.name]["x_err"]]
        if refs and not e <= K * max(refs) + FLOOR:
            bad.append(f"{label}: iterate error {e:.3e} at iteration {k} vs the double oracles' {refs}")
    kf = min([p["iterations_in_common"]] + [cmps[t.name]["iterations_in_common"] for t in doubles])
    if kf > 0:
        # The double oracles share libm with the arbiter; the product evaluates ∇f with its own exp / division (each within
        # an ulp), so near convergence — ‖g_k‖ ≪ ‖g_1‖ — its ‖g‖ differs by evaluation noise that no summation order has:
        # an absolute floor of 1e-13·‖g_1‖ on ‖g‖ (and of 1e-13·|f_1| on f), on top of K × the oracles' drift.
        fl_f = 1e-13 * abs(arb.f[0]) / np.maximum(np.abs(arb.f[:kf]), 1e-300) + 1e-12
        fl_g = 1e-13 * abs(arb.gnorm[0]) / np.maximum(np.abs(arb.gnorm[:kf]), 1e-300) + 1e-12
        ref_f = np.max([cmps[t.name]["f_err"][:kf] for t in doubles], axis=0)
        ref_g = np.max([cmps[t.name]["g_err"][:kf] for t in doubles], axis=0)
        # running maxima: an error curve is noisy iteration by iteration, its envelope is what amplification produces
        for nm, mine, ref, fl in (("f", p["f_err"][:kf], ref_f, fl_f), ("‖g‖", p["g_err"][:kf], ref_g, fl_g)):
            over = np.maximum.accumulate(mine) > K * np.maximum.accumulate(ref) + np.maximum.accumulate(fl)
            if over.any():
                i = int(np.argmax(over))
                bad.append(f"{label}: {nm} error envelope {np.maximum.accumulate(mine)[i]:.3e} at iteration {i + 1} vs the oracles' "
                           f"{np.maximum.accumulate(ref)[i]:.3e} (floor {np.maximum.accumulate(fl)[i]:.1e})")
    rec["violations"] = bad
    return rec, bad


# ------------------------------------------------------------------ the arbiter's own pins (CPU tier)
def test_exact_sums_equal_float128_accumulation():
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r)
from oracle import oracle as O
O.use_exact(True)
L = O.lib()
assert L.orc_exact_sums() == 1
rng = np.random.default_rng(7)
bad = 0
for n in (1, 2, 7, 64, 1000, 65535, 65536, 65537, 300001):
    for scale in (0.0, 8.0, 30.0):     # graded magnitudes: up to e^±30
        a = rng.standard_normal(n) * np.exp(rng.uniform(-scale, scale, n))
        b = rng.standard_normal(n) * np.exp(rng.uniform(-scale, scale, n))
        d, q = L.orc_dot(O._dp(a), O._dp(b), n), L.orc_dot_f128(O._dp(a), O._dp(b), n)
        s, sq = L.orc_sum(O._dp(a), n), L.orc_sum_f128(O._dp(a), n)
        bad += (d != q) + (s != sq)
# a cancelling sum (condition number ~ 1e12): still the correctly rounded value
m = 100000
a = rng.standard_normal(m); b = rng.standard_normal(m)
a2 = np.concatenate([a, a]); b2 = np.concatenate([b, -b * (1 + 1e-12)])
d, q = L.orc_dot(O._dp(a2), O._dp(b2), 2 * m), L.orc_dot_f128(O._dp(a2), O._dp(b2), 2 * m)
bad += (d != q)
print(bad)
""" % ROOT
    outs = []
    for threads in ("1", "4"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS=threads), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-800:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs == ["0", "0"], outs


def test_arbiter_does_not_depend_on_the_thread_count(tmp_path):
    a = oracle_child("c3", 200_000, 30, "exact", [10, 30], tmp_path, 1)
    os.rename(str(tmp_path / "lh_c3_exact.npz"), str(tmp_path / "one.npz"))
    b = oracle_child("c3", 200_000, 30, "exact", [10, 30], tmp_path, 4)
    assert np.array_equal(a.log_a, b.log_a) and np.array_equal(a.f, b.f) and np.array_equal(a.snap_x, b.snap_x)
    assert np.array_equal(a.log_margin, b.log_margin) and np.all(a.log_margin >= 0.0)


def test_arbiter_reproduces_the_first_booth_line_search():
    """SURVEY.md appendix A.2 (hand-derived): a* = 0.0625 after five evaluations at 1, 1/2, 1/4, 1/8, 1/16 — and the
    margins of those decisions are the hand-computable ones (ϕ(1) = 7121.7378 against ϕ₀ + c1·a·dϕ₀ = 25.3513…)."""
    code = r"""
import numpy as np, sys, json
sys.path.insert(0, %r)
from oracle import oracle as O
O.use_exact(True)
r = O.minimizeobjective(O.objective("booth"), np.array([0.43, 1.23]), O.cg_config(1e-5, O.beta_config("HagerZhang"), 1000),
                        O.strong_wolfe(1e-5, 0.8), log_cap=4096, snap_iters=[1])
print(json.dumps(dict(status=r.status, a=r.log_a[:5].tolist(), m=r.log_margin[:5].tolist(), x=r.minimizer.tolist(),
                      step=float(r.trace_step_size[0]), evals=int(r.trace_objective_evals[0]), snap=r.snap_x[0].tolist())))
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["status"] == "success" and d["a"] == [1.0, 0.5, 0.25, 0.125, 0.0625] and d["step"] == 0.0625 and d["evals"] == 5
    assert np.allclose(d["x"], [1.0, 3.0], atol=1e-5)
    # first evaluation: ϕ(1) = 7121.7378 > 25.3602 + 1e-5·1·(−889.9272): margin |7121.7378 − 25.3513…| / 7121.7378
    assert abs(d["m"][0] - (7121.7378 - (25.3602 - 1e-5 * 889.9272)) / 7121.7378) < 1e-12
    # x after iteration 1 = x0 + 0.0625·(19.86, 22.26)
    assert np.allclose(d["snap"], [0.43 + 0.0625 * 19.86, 1.23 + 0.0625 * 22.26], rtol=0, atol=1e-15)


# ------------------------------------------------------------------ CPU tier: the product's host engine over the test double
CPU_CASES = {"c2": (100_003 - 1, 200), "c3": (100_000, 60), "c4": (60_000, 40), "c5": (400_000, 100)}


@pytest.mark.parametrize("config", sorted(CPU_CASES))
def test_long_horizon_host_engine_vs_arbiter(cgo, tmp_path, config):
    n, iters = CPU_CASES[config]
    ck = list(range(20, iters + 1, 20))
    arb = oracle_child(config, n, iters, "exact", ck, tmp_path, 2)
    doubles = [oracle_child(config, n, iters, "c", ck, tmp_path, 1), oracle_child(config, n, iters, "omp", ck, tmp_path, 3)]
    # the host engine has no checkpoint hook: one run per checkpoint (small n), the last one carries the logs
    snaps, last = [], None
    for k in ck:
        last = run_hostsim(baseline_case(config, n, k), points=1 if config == "c4" else 3)
        snaps.append(last.minimizer.copy())
        if last.iters_ran < k:      # the solve ended before this checkpoint: nothing further to compare
            snaps.pop()
            break
    full = last if last.iters_ran >= iters or len(snaps) < len(ck) else run_hostsim(baseline_case(config, n, iters), points=1 if config == "c4" else 3)
    prod = LH.Traj("host_engine", full.log_a, full.trace_objective_evals, full.trace_objective, full.trace_grad_norm, full.trace_step_size,
                   full.status, full.iters_ran, np.array(ck[:len(snaps)], dtype=np.int64), np.array(snaps))
    rec, bad = check(prod, doubles, arb, f"{config} n={n}", K_ITERATE[config])
    print("\n" + json.dumps({config: rec["host_engine"]}))
    assert not bad, bad
    assert rec["host_engine"]["iterations_in_common"] >= 10


# ------------------------------------------------------------------ GPU tier: the BASELINE sizes over the horizon bench.py times
GPU_CASES = {"c2": (10**6, 210), "c3": (10**7, 210), "c4": (10**7, 100), "c5": (10**8, 120)}


def run_gpu_traj(cgo, c, ctx, ckpts):
    from _cases import _product_structs, gpu_objective
    _, _lib, cfg, ls = _product_structs(c)
    obj = gpu_objective(c, ctx)
    s = cgo.Solver(obj, cfg, ls)
    snaps, reached = [], []
    try:
        s.enable_trial_log()
        s.set_x0(c.x0)
        s.start()
        done, fin = 0, False
        for k in list(ckpts) + [c.max_iters]:
            if k > done and not fin:
                fin = s.iterate(k - done)
                done = k
            r = s.results(vectors=(k in ckpts))
            if k in ckpts and r.iters_ran >= k and len(reached) < len(ckpts) and k not in reached:
                snaps.append(r.minimizer.copy()); reached.append(k)
        r = s.results(vectors=False)
        la, _, _ = s.trial_log()
        fam, sym = s.kernel_family(), s.kernel_symbol("accept_dir_trial")
    finally:
        s.close(); obj.close()
    return LH.Traj("gpu", la, r.trace.objective_evals, r.trace.objective, r.trace.grad_norm, r.trace.step_size, r.status, r.iters_ran,
                   np.array(reached, dtype=np.int64), np.array(snaps)), fam, sym


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["c2", "c3", "c4", "c5"])
def test_long_horizon_at_baseline_size_vs_arbiter(cgo, gpu_ctx, tmp_path, config):
    n, iters = GPU_CASES[config]
    if os.environ.get("CGO_TEST_LONG") == "1" and config == "c5":
        iters = 255                                  # bench.py's default run: 5 warm-up + 5 windows of 50
    ck = [k for k in (10, 25, 50, 100, 150, 200, 250) if k <= iters]
    threads = _cores()
    arb = oracle_child(config, n, iters, "exact", ck, tmp_path, threads)
    doubles = [oracle_child(config, n, iters, "omp", ck, tmp_path, threads)]
    if n <= 10**7:
        doubles.append(oracle_child(config, n, iters, "c", ck, tmp_path, 1))
    prod, fam, sym = run_gpu_traj(cgo, baseline_case(config, n, iters), gpu_ctx, ck)
    rec, bad = check(prod, doubles, arb, f"{config} n={n:.0e}", K_ITERATE[config])
    rec["_facts"] = dict(config=config, n=n, iters=iters, kernel_family=fam, kernel=sym, oracle_threads=threads, library_build_id=cgo.build_id())
    print("\n" + json.dumps(rec))
    out = os.environ.get("CGO_LONG_HORIZON_OUT")
    if out:
        os.makedirs(out, exist_ok=True)
        json.dump(rec, open(os.path.join(out, f"long_horizon_{config}.json"), "w"), indent=1)
    assert not bad, bad
    # the statement must cover a real stretch of what bench.py times
    assert rec["gpu"]["iterations_in_common"] >= min(20, arb.iters_ran)


@pytest.mark.gpu
def test_long_horizon_sharded_over_eight_virtual_ranks_vs_arbiter(cgo, gpu_ctx, tmp_path):
    """Config 5's 8-GPU layout over the bench's horizon: EIGHT contexts in one process on this GPU, each a rank with its
    contiguous shard (n = 2e7: 2.5e6 elements per rank), exchanging one block per launch through the cgo_allgather_fn ABI in lock
    step and merging in rank order — 120 iterations against the arbiter, by the same three criteria as the unsharded runs.
    (The sums of a sharded run are eight partial sums added in rank order: another valid summation order, so its trajectory
    is its own; what must hold is that it stays on the arbiter's branch where the arbiter is decided, and as close to it as
    the double oracles are.)  Every rank must report the same scalars bit for bit throughout."""
    import threading
    config, n, iters, W = "c5", 2 * 10**7, 120, 8
    ck = [10, 25, 50, 100]
    threads = _cores()
    arb = oracle_child(config, n, iters, "exact", ck, tmp_path, threads)
    doubles = [oracle_child(config, n, iters, "omp", ck, tmp_path, threads), oracle_child(config, n, iters, "c", ck, tmp_path, 1)]
    c = baseline_case(config, n, iters)
    bar = threading.Barrier(W)
    slots, outs, errs = [None] * W, [None] * W, []

    def make_allgather(rank):
        def ag(send):
            slots[rank] = send.copy()
            bar.wait()
            out = np.concatenate(slots)
            bar.wait()
            return out
        return ag

    def worker(rank):
        try:
            ctx = cgo.Context(0)
            ctx.set_comm_callback(rank, W, make_allgather(rank))
            outs[rank] = run_gpu_traj(cgo, c, ctx, ck)[0]
            ctx.close()
        except Exception as e:  # pragma: no cover
            errs.append(e)
            bar.abort()
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    t0 = outs[0]
    for o in outs[1:]:
        assert o.iters_ran == t0.iters_ran and o.status == t0.status
        assert np.array_equal(o.log_a, t0.log_a) and np.array_equal(o.f, t0.f) and np.array_equal(o.gnorm, t0.gnorm)
        assert np.array_equal(o.snap_iters, t0.snap_iters)
    whole = LH.Traj("gpu", t0.log_a, t0.evals, t0.f, t0.gnorm, t0.step, t0.status, t0.iters_ran, t0.snap_iters,
                    np.concatenate([o.snap_x for o in outs], axis=1))
    rec, bad = check(whole, doubles, arb, f"{config} n={n:.0e} over {W} virtual ranks", K_ITERATE[config])
    print("\n" + json.dumps(rec))
    assert not bad, bad
    assert rec["gpu"]["iterations_in_common"] >= min(20, arb.iters_ran)
