import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cgo_amd as cgo
from _cases import O
n = 100000
D = O.fill_uniform(n, 6, 1.0, 10.0)
cfg = cgo.setupCGConfig(1e-5, cgo.HagerZhang(), cgo.EnableTrace(), max_iters=500)
ls = cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50)
def run():
    return cgo.primalbarriermethod(cgo.BoxConstraints(0.5, 4.0), "ObjQuadDiag", np.ones(n), cfg, ls,
                                   cgo.setupPrimalBarrierConfig(1e-3, 10.0, 12, t_initial=1.0), param=D)
os.environ["CGO_MULTI_MIN_N"] = "9000000000000000000"
a = run()
del os.environ["CGO_MULTI_MIN_N"]
b = run()
for k in range(2):
    ta, tb = a.centering_results[k][-1].trace, b.centering_results[k][-1].trace
    m = min(len(ta.step_size), len(tb.step_size))
    sa, sb = np.asarray(ta.step_size[:m]), np.asarray(tb.step_size[:m])
    fa, fb = np.asarray(ta.objective[:m]), np.asarray(tb.objective[:m])
    d = np.nonzero(sa != sb)[0]
    print("centering", k, "lens", len(ta.step_size), len(tb.step_size), "first step divergence", d[:1], "max rel f diff before", (np.max(np.abs(fa[:d[0]] - fb[:d[0]]) / np.abs(fa[:d[0]])) if len(d) else np.max(np.abs(fa - fb) / np.abs(fa))))
    if len(d):
        i = d[0]
        print("  around:", sa[max(0, i - 2):i + 3], sb[max(0, i - 2):i + 3], fa[i - 1:i + 2], fb[i - 1:i + 2], np.asarray(ta.objective_evals[i - 1:i + 2]), np.asarray(tb.objective_evals[i - 1:i + 2]))
