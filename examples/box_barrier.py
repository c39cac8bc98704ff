#!/usr/bin/env python3
"""Box-constrained minimisation with the primal barrier method on the GPU (the problem of the reference's
examples/constrained.jl: Booth inside [−10, 10]²), plus a 10⁶-variable quadratic the reference's dense
2D×D constraint Jacobian could not hold.

    python examples/box_barrier.py        # needs an MI355X
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import cgo_amd as cgo

cfg = cgo.setupCGConfig(1e-5, cgo.HagerZhang(), cgo.EnableTrace(), max_iters=1000)
wolfe = cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50)
armijo = cgo.Backtracking(cgo.Armijo(1e-3), 0.9, 300, 50)
fallback = cgo.setupCGConfig(1e-5, cgo.LiuStorrey(), cgo.EnableTrace(), max_iters=1000)
dfp = cgo.setupCGConfig(1e-5, cgo.setupBroydenFamily(1.0, 2), cgo.EnableTrace(), max_iters=1000)   # Broyden DFP, as in the reference's script

res = cgo.primalbarriermethod(cgo.BoxConstraints(-10.0, 10.0), "ObjBooth", [0.43, 1.23], cfg, wolfe,
                              cgo.setupPrimalBarrierConfig(1e-8, 10.0, 100), (dfp, armijo), (fallback, wolfe))
good = [c[-1] for c in res.centering_results if c[-1].status == "success"]
print("Booth in [-10,10]^2:", res.status, "after", res.iters_ran, "centering steps, t_final =", res.t_final)
print("  last centre:", good[-1].minimizer, " objective evaluations:", res.total_objective_evals)

n = 1_000_000
D = 1.0 + 9.0 * np.random.default_rng(24).random(n)
res = cgo.primalbarriermethod(cgo.BoxConstraints(0.5, 4.0), "ObjQuadDiag", np.ones(n), cfg, wolfe,
                              cgo.setupPrimalBarrierConfig(1e-3, 10.0, 6, t_initial=1.0), param=D)
good = [c[-1] for c in res.centering_results if c[-1].status == "success"]
print(f"quadratic, n = {n}: {res.status} after {res.iters_ran} centering steps; "
      f"min x = {good[-1].minimizer.min():.6f} (lower bound 0.5)")
