#!/usr/bin/env python3
"""Minimise the Booth function with Hager–Zhang CG + the strong-Wolfe bisection line search on the GPU —
the workflow of the reference's examples/min.jl (configuration values from there), through `cgo_amd`.

    python examples/booth_min.py          # needs an MI355X
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import cgo_amd as cgo

fdf = cgo.Booth()                                  # device-side objective descriptor (2 variables)
x0 = np.array([0.43, 1.23])

ls = cgo.setupStrongWolfeBisection(1e-5, 0.8, a_max_growth_factor=2.0, max_iters=1000, zoom_max_iters=100)
cfg = cgo.setupCGConfig(1e-5, cgo.HagerZhang(), cgo.EnableTrace(), max_iters=1000)

g0 = np.empty(2)
print("f(x0) =", fdf(g0, x0), " ∇f(x0) =", g0)     # the fdf!(g, x) contract, evaluated on the device

ret = cgo.minimizeobjective(fdf, x0, cfg, ls)
print("status       :", ret.status)
print("minimizer    :", ret.minimizer, "(global minimum: [1, 3])")
print("objective    :", ret.objective)
print("‖gradient‖   :", float(np.linalg.norm(ret.gradient)))
print("iterations   :", ret.iters_ran, " objective evaluations:", int(ret.trace.objective_evals.sum()))
assert ret.status == "success" and np.allclose(ret.minimizer, [1.0, 3.0], atol=1e-4)

# fall-back chain: rerun from the last iterate with another configuration if the first does not succeed
rets = cgo.minimizeobjectivererun(fdf, x0, cgo.setupCGConfig(1e-5, cgo.HagerZhang(), cgo.EnableTrace(), max_iters=3), ls,
                                  (cgo.setupCGConfig(1e-5, cgo.LiuStorrey(), cgo.EnableTrace(), max_iters=1000),
                                   cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50)))
print("rerun chain  :", [(r.status, r.iters_ran) for r in rets])
