#!/usr/bin/env python3
"""L-BFGS (the new QNβConfig, m = 10) on the log-sum-exp objective of BASELINE config 4,
f(x) = log Σ exp(x_i) + ½λ‖x‖², through the reference's own call form:

    ret = minimizeobjective(fdf, x_initial, config, linesearch_config)

On the device an outer iteration is ONE launch: the direction pass also evaluates the first trial of the next line
search and every inner product the next state update needs; `Solver.lbfgs_stats()` says how the
state updates were paid for.

    python examples/lse_lbfgs.py [n]      # needs an MI355X
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import cgo_amd as cgo

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
lam = 1e-2 / n
x0 = 5.0 * (2.0 * np.random.default_rng(24).random(n) - 1.0)
config = cgo.setupCGConfig(1e-9, cgo.LBFGS(10), cgo.EnableTrace(), max_iters=60)
ls = cgo.setupStrongWolfeBisection(1e-5, 0.9)

# the one-call form
ret = cgo.minimizeobjective(cgo.LogSumExp(n, lam), x0, config, ls)
print(f"n = {n}: {ret.status} after {ret.iters_ran} iterations, f = {ret.objective:.12g}, ‖g‖ = {ret.trace.grad_norm[-1]:.3e}, "
      f"{int(np.sum(ret.trace.objective_evals))} objective evaluations in {ret.total_launches} launches")

# the same solve through a Solver handle, to look at how the pushes were paid for
obj = cgo.LogSumExp(n, lam)
s = cgo.Solver(obj, config, ls)
s.set_x0(x0)
s.start()
t = time.perf_counter()
while not s.iterate(1 << 30):
    pass
r = s.results(vectors=False)
dt = time.perf_counter() - t
sp, fu, pl = s.lbfgs_stats()
print(f"  {r.iters_ran / dt:.0f} iterations/s; state updates: {sp} speculated by a direction pass, {fu} fused pushes, {pl} plain")
s.close(); obj.close()
