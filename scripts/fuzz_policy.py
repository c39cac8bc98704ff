#!/usr/bin/env python3
"""Seeded sweep over launch POLICIES (cgo_solver_policy): random small problems × random policy combinations, each held to the
default-policy run of the same problem (same step sequence, status, iterate to 1e-10 — a policy never changes what is
computed) and to the oracle.  python3 scripts/fuzz_policy.py [count] [seed]  — prints every failing case."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
import cgo_amd as cgo
from _cases import Case, O, assert_parity, quad_D, run_oracle, first_divergence, rel
from _suite import rosen_x0
from test_policy import run

count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
for k in ("CGO_RESIDENT", "CGO_CTL_DEPTH", "CGO_MULTI_MIN_N", "CGO_MULTI5_MIN_N", "CGO_MULTI7_MIN_N", "CGO_BIG_BYTES", "CGO_TAIL_STRICT", "CGO_FUSED_TAIL"):
    os.environ.pop(k, None)
bad = 0


def pick(*xs):
    return xs[int(rng.integers(0, len(xs)))]


for k in range(count):
    n = int(rng.integers(2, 40000))
    beta = pick("PolakRibiere", "HagerZhang", "DaiYuan", "HestenesStiefel", "LiuStorrey", "YuanWangSheng", "SallehAlhawarat")
    wolfe = bool(rng.integers(0, 2))
    kw = dict(beta=beta, max_iters=int(rng.integers(4, 16)), eps=1e-9)
    if wolfe:
        kw.update(ls="WolfeBisection", cond="Wolfe", c1=1e-3, c2=0.9, ls_max_iters=100)
    else:
        kw.update(c2=0.1 if beta == "PolakRibiere" else pick(0.1, 0.5, 0.8))
    if rng.integers(0, 3) < 2:
        c = Case(f"pz{k}-q{n}-{beta}", "quad_diag", n, O.fill_uniform(n, 900 + k, -2.0, 2.0), D=quad_D(n, 1.0, pick(2.0, 50.0, 1000.0), seed=1300 + k), **kw)
        tol = 1e-10
    else:
        n += n & 1
        kw["max_iters"] = min(kw["max_iters"], 6); kw["eps"] = 1e-12
        c = Case(f"pz{k}-r{n}-{beta}", "rosenbrock_paired", n, rosen_x0(n, 0.05, 77 + k), **kw)
        tol = 1e-8
    pol = dict(points=pick(None, 1, 3, 5, 7), resident=pick(None, False, True), controller_depth=pick(None, 0, 2, 8),
               controller_fused=pick(None, False), fused_tail=pick(None, False), strict_tail=pick(None, None, True),
               hbm_stream_bytes=pick(None, None, 1.0), resident_points=pick(None, 1, 3, 7), resident_chunk=pick(None, 512, 2048),
               stored_gradient=pick(None, None, None, True), controller_graph=pick(None, None, True))
    pol = {a: b for a, b in pol.items() if b is not None}
    name = c.name + " " + " ".join(f"{a}={b}" for a, b in pol.items())
    try:
        ref = run_oracle(c)
        ctx = cgo.Context(0)
        try:
            base, _ = run(cgo, c, None, ctx)
        finally:
            ctx.close()
        ctx = cgo.Context(0)
        try:
            got, facts = run(cgo, c, cgo.SolverPolicy(**pol), ctx)
        finally:
            ctx.close()
        assert first_divergence(got, base) is None and got.status == base.status and got.iters_ran == base.iters_ran, "step sequence differs from the default policy's"
        assert rel(got.minimizer, base.minimizer) <= 1e-10, f"iterate differs from the default policy's by {rel(got.minimizer, base.minimizer):.2e}"
        assert_parity(got, ref, tol, name)
    except AssertionError as e:
        bad += 1
        print("FAIL", name, "::", str(e)[:300], flush=True)
    except Exception as e:  # an API error is a failure too
        bad += 1
        print("ERROR", name, "::", repr(e)[:300], flush=True)
    if (k + 1) % 50 == 0:
        print(f"  {k + 1} cases, {bad} failed so far", flush=True)
print(f"{count} cases, {bad} failed")
sys.exit(1 if bad else 0)
