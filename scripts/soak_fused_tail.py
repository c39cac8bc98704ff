#!/usr/bin/env python3
"""Soak of the fence-free reduction tail under GPU sharing: W processes on ONE GPU run long solves at the same time (their
launches interleave and pre-empt one another), each with launches that finish their own sums (default) and again with the
finalize launches (CGO_FUSED_TAIL=0); the two must agree bit for bit, iteration count for iteration count.
usage: soak_fused_tail.py [workers] [iterations]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cgo_amd as cgo
def solve(ctx, kind, n, iters):
    if kind == "quad":
        obj = cgo.QuadDiagRandom(n, 24 + RANK, 1.0, 1000.0, ctx)
        cfg = cgo.setupCGConfig(1e-300, cgo.PolakRibiere(), cgo.EnableTrace(), max_iters=iters)
        ls = cgo.setupStrongWolfeBisection(1e-5, 0.1)
        fill = ("constant", 1.0)
    else:
        obj = cgo.RosenbrockPaired(n, ctx)
        cfg = cgo.setupCGConfig(1e-300, cgo.HagerZhang(), cgo.EnableTrace(), max_iters=iters)
        ls = cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50)
        fill = ("alternate", -1.2, 1.0)
    s = cgo.Solver(obj, cfg, ls)
    s.set_x0_fill(*fill)
    s.start()
    while not s.iterate(1 << 40):
        pass
    r = s.results()
    s.close(); obj.close()
    return r
fused = cgo.Context(0)
os.environ["CGO_FUSED_TAIL"] = "0"
plain = cgo.Context(0)
del os.environ["CGO_FUSED_TAIL"]
bad = 0
launches = 0
for kind, n in (("quad", 1 << 20), ("quad", 40000), ("rosen", 200000), ("quad", 3 << 20), ("rosen", 4096)):
    a, b = solve(fused, kind, n, ITERS), solve(plain, kind, n, ITERS)
    same = (a.status == b.status and a.iters_ran == b.iters_ran and a.objective == b.objective
            and np.array_equal(a.minimizer, b.minimizer) and np.array_equal(a.trace.objective, b.trace.objective, equal_nan=True))
    launches += a.total_launches
    print(f"rank {RANK} {kind} n={n}: {a.status} after {a.iters_ran} iterations, {a.total_launches} launches, f={a.objective!r} {'==' if same else '!='} finalize-launch run", flush=True)
    bad += 0 if same else 1
print(f"rank {RANK}: {launches} fused launches, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
"""
def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    procs = []
    for r in range(w):
        code = f"ROOT={ROOT!r}\nRANK={r}\nITERS={iters}\n" + WORKER
        procs.append(subprocess.Popen([sys.executable, "-c", code]))
    rc = 0
    for p in procs:
        rc |= p.wait()
    print("soak:", "FAILED" if rc else "ok")
    sys.exit(rc)
if __name__ == "__main__":
    main()
