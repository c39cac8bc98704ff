"""What the first centering steps of tests/test_primal_barrier.py::test_primalbarriermethod_large_elementwise end in, per launch policy."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cgo_amd as cgo
from oracle import oracle as O
n = 100000
D = O.fill_uniform(n, 6, 1.0, 10.0)
cfg = cgo.setupCGConfig(1e-5, cgo.HagerZhang(), cgo.EnableTrace(), max_iters=500)
ls = cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50)
BIGN = "9000000000000000000"
for name, env in (("default (resident)", {}), ("host-driven, 3 points", {"CGO_RESIDENT": "0", "CGO_MULTI_MIN_N": "0", "CGO_MULTI5_MIN_N": BIGN, "CGO_MULTI7_MIN_N": BIGN}),
                  ("host-driven, 1 point", {"CGO_RESIDENT": "0", "CGO_MULTI_MIN_N": BIGN, "CGO_MULTI5_MIN_N": BIGN, "CGO_MULTI7_MIN_N": BIGN})):
    for k in ("CGO_RESIDENT", "CGO_MULTI_MIN_N", "CGO_MULTI5_MIN_N", "CGO_MULTI7_MIN_N"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for rep in range(2):
        r = cgo.primalbarriermethod(cgo.BoxConstraints(0.5, 4.0), "ObjQuadDiag", np.ones(n), cfg, ls,
                                    cgo.setupPrimalBarrierConfig(1e-3, 10.0, 12, t_initial=1.0), param=D)
        print(name, "|", r.status, r.iters_ran, [(rr[-1].status, rr[-1].iters_ran) for rr in r.centering_results[:4]], flush=True)
