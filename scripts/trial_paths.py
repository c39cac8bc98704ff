#!/usr/bin/env python3
"""Which steps do the line searches of the headline workload ask for, iteration by iteration, relative to the first
step a0 of each search — and which of them did the speculative launch carry?  (input for the choice of the six
speculative points, cgo_ctl.hpp ls_trial_points_n)   usage: trial_paths.py [n] [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cgo_amd as cgo

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
kind = sys.argv[3] if len(sys.argv) > 3 else "quad"      # quad | rosen (paired Rosenbrock, config 1) | chain
ctx = cgo.default_context()
cfg = cgo.setupCGConfig(1e-30, cgo.PolakRibiere(), cgo.EnableTrace(), max_iters=iters)
if kind == "quad":
    obj = cgo.QuadDiagRandom(n, 24, 1.0, 1000.0, ctx)
elif kind == "rosen":
    obj = cgo.RosenbrockPaired(n, ctx)
else:
    obj = cgo.RosenbrockChained(n, ctx)
s = cgo.Solver(obj, cfg, cgo.setupStrongWolfeBisection(1e-5, 0.1))
s.enable_trial_log()
if kind == "quad":
    s.set_x0_fill("constant", 1.0)
else:
    s.set_x0_fill("alternate", -1.2, 1.0)
s.start()
seen, launches = 0, s.results(vectors=False).total_launches
for it in range(iters):
    done = s.iterate(1)
    la, lp, ld = s.trial_log()
    r = s.results(vectors=False)
    steps = list(la[seen:])
    seen = len(la)
    dl = r.total_launches - launches
    launches = r.total_launches
    if steps:
        a0 = steps[0]
        print(f"it {it + 1:3d} launches {dl} a0 {a0:.6g} path " + " ".join(f"{a / a0:.6g}" for a in steps), flush=True)
    if done:
        break
s.close(); obj.close()
