#!/usr/bin/env python3
"""Ad-hoc sweep of L-BFGS on log-sum-exp at EXTREME ranges (|x0| up to 1e4: exp overflow / underflow against any fixed
reference; λ from 0 to 100): the one-pass form must come out as the two-pass form does — same status, same step sequence,
same iterates — and as the oracle does."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from _cases import Case, O, run_gpu, run_oracle, first_divergence, rel
count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)
bad = 0
for k in range(count):
    n = int(rng.integers(1, 2500))
    scale = float(rng.choice([100.0, 700.0, 1e4]))
    lam = float(rng.choice([0.0, 1e-12, 1e-4, 1.0, 100.0]))
    m = int(rng.integers(1, 11))
    wolfe = bool(rng.integers(0, 2))
    kw = dict(beta="LBFGS", m=m, max_iters=int(rng.integers(3, 10)), eps=1e-5)
    if wolfe:
        kw.update(ls="WolfeBisection", c1=1e-3, c2=0.9, ls_max_iters=100)
    else:
        kw.update(c2=float(rng.choice([0.1, 0.9])))
    c = Case(f"x{k}-n{n}-s{scale:g}-l{lam:g}-m{m}-{'wb' if wolfe else 'sw'}", "lse", n, scale * O.fill_uniform(n, 3000 + k, -1.0, 1.0), lam=lam, **kw)
    try:
        ref = run_oracle(c)
        os.environ.pop("CGO_LBFGS_SPEC", None)
        one = run_gpu(c)
        os.environ["CGO_LBFGS_SPEC"] = "0"
        two = run_gpu(c)
        os.environ.pop("CGO_LBFGS_SPEC")
        assert one.status == two.status and one.iters_ran == two.iters_ran and first_divergence(one, two) is None, ("one vs two", one.status, two.status, one.iters_ran, two.iters_ran, first_divergence(one, two))
        if one.iters_ran:
            assert rel(one.minimizer, two.minimizer) <= 1e-9, ("one vs two x", rel(one.minimizer, two.minimizer))
        assert one.status == ref.status and one.iters_ran == ref.iters_ran and first_divergence(one, ref) is None, ("vs oracle", one.status, ref.status, one.iters_ran, ref.iters_ran, first_divergence(one, ref))
        if one.iters_ran:
            assert rel(one.minimizer, ref.minimizer) <= 1e-9, ("vs oracle x", rel(one.minimizer, ref.minimizer))
    except AssertionError as e:
        bad += 1
        print("FAIL", c.name, str(e)[:260], flush=True)
print(f"{count} cases, {bad} failed")
