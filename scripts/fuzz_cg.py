#!/usr/bin/env python3
"""Ad-hoc seeded sweep of the CG engine (k_cg family under its default policy: resident where it applies, launches otherwise;
the stored-gradient family through YuanWeiLu etc.) against the oracle: quadratic (well conditioned: a divergence here is a
bug, not chaos), short horizons.  python3 scripts/fuzz_cg.py [count] [seed]  — prints every case that fails parity."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from _cases import Case, O, assert_parity, quad_D, run_gpu, run_oracle, first_divergence, rel
from _suite import BETAS
count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
bad = 0
for k in range(count):
    n = int(rng.integers(1, 6000))
    beta = str(rng.choice(BETAS))
    lsk = int(rng.integers(0, 4))
    kw = dict(beta=beta, max_iters=int(rng.integers(4, 14)), eps=1e-9)
    if lsk == 0:
        kw.update(c2=0.1 if beta == "PolakRibiere" else float(rng.choice([0.1, 0.5, 0.8])))
    elif lsk == 1:
        kw.update(ls="WolfeBisection", cond="Wolfe", c1=1e-3, c2=0.9, ls_max_iters=100)
    elif lsk == 2:
        kw.update(ls="WolfeBisection", cond="YuanWeiLuWolfe", c1=1e-3, c2=0.9, delta1=1e-4, ls_max_iters=100)
    else:
        kw.update(ls="Backtracking", c1=1e-3, discount=float(rng.choice([0.5, 0.9])), ls_max_iters=100)
    hi = float(rng.choice([2.0, 50.0, 1000.0]))
    which = int(rng.integers(0, 4))
    if which <= 1:
        c = Case(f"cg{k}-n{n}-{beta}-ls{lsk}-hi{hi:g}", "quad_diag", n, O.fill_uniform(n, 900 + k, -2.0, 2.0), D=quad_D(n, 1.0, hi, seed=1300 + k), **kw)
    else:   # the two Rosenbrock forms: short horizons (chaotic beyond)
        from _suite import rosen_x0
        kw["max_iters"] = min(kw["max_iters"], 5); kw["eps"] = 1e-12
        n2 = n + (n & 1) if which == 2 else max(n, 3)
        c = Case(f"cg{k}-{'rp' if which == 2 else 'rc'}{n2}-{beta}-ls{lsk}", "rosenbrock_paired" if which == 2 else "rosenbrock_chained", n2, rosen_x0(n2 + (n2 & 1), 0.05, 77 + k)[:n2], **kw)
    try:
        ref = run_oracle(c)
        for res in ("1", "0"):
            os.environ["CGO_RESIDENT"] = res
            got = run_gpu(c)
            assert_parity(got, ref, 1e-10 if c.objective == 'quad_diag' else 1e-8, c.name + f" resident={res}", step_rtol=1e-12 if lsk == 3 else 0.0)
    except AssertionError as e:
        bad += 1
        print("FAIL", str(e)[:300], flush=True)
print(f"{count} cases, {bad} failed")
