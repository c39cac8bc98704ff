#!/usr/bin/env python3
"""Ad-hoc sweep: L-BFGS (one ring pass per iteration) SHARDED over W = 2 … 5 virtual ranks on one GPU (W contexts in one
process exchanging through the cgo_allgather_fn ABI) against the unsharded run — three objectives, ragged shard sizes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import cgo_amd as cgo
import test_gpu_parity as T
from _cases import Case, O, quad_D
from _suite import rosen_x0
count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 555)
bad = 0
for k in range(count):
    W = int(rng.integers(2, 6))
    n = int(rng.integers(2 * W + 2, 4000))
    m = int(rng.integers(1, 11))
    obj = ("lse", "quad_diag", "rosenbrock_paired")[k % 3]
    kw = dict(beta="LBFGS", m=m, max_iters=int(rng.integers(4, 10)), eps=1e-5 if obj != "rosenbrock_paired" else 1e-12, c2=float(rng.choice([0.1, 0.9])))
    if obj == "lse":
        c = Case(f"sh{k}-lse{n}-W{W}-m{m}", "lse", n, float(rng.choice([0.5, 5.0, 30.0])) * O.fill_uniform(n, 700 + k, -1.0, 1.0), lam=float(rng.choice([1e-6, 1e-3, 1e-1])), **kw)
    elif obj == "quad_diag":
        c = Case(f"sh{k}-quad{n}-W{W}-m{m}", "quad_diag", n, O.fill_uniform(n, 800 + k, -2.0, 2.0), D=quad_D(n, 1.0, 100.0, seed=900 + k), **kw)
    else:
        n += n & 1
        kw["max_iters"] = min(kw["max_iters"], 6)
        c = Case(f"sh{k}-rosen{n}-W{W}-m{m}", "rosenbrock_paired", n, rosen_x0(n, 0.05, 60 + k), **kw)
    try:
        T._two_virtual_ranks(cgo, c, W=W)
    except AssertionError as e:
        bad += 1
        print("FAIL", c.name, str(e)[:300], flush=True)
print(f"{count} cases, {bad} failed")
