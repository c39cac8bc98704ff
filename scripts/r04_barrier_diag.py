"""Diagnostic (GPU box): where do the GPU's centering steps and the numpy restatement's part?  
Prints per centering step and stage: statuses, iteration counts, first step-size divergence, minimizer distance —
for the device barrier objective (device `log`) and for the closure path (numpy evalbarrier: libm `log`)."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cgo_amd as cgo                                     # noqa: E402
from _cases import N                                       # noqa: E402
from test_primal_barrier import X0, _oracle_run            # noqa: E402


def first_div(a, b, tol=1e-9):
    m = min(len(a), len(b))
    for i in range(m):
        if abs(a[i] - b[i]) > tol * max(abs(a[i]), abs(b[i]), 1e-300):
            return i
    return None if len(a) == len(b) else m


def report(tag, got_rets, ref):
    print(f"== {tag}")
    for k, (gr, rr) in enumerate(zip(got_rets, ref.centering_results)):
        for j in range(max(len(gr), len(rr))):
            if j >= len(gr) or j >= len(rr):
                print(f"  centering {k} stage {j}: only in {'gpu' if j < len(gr) else 'ref'}")
                continue
            a, b = gr[j], rr[j]
            fd = first_div(list(a.trace.step_size[:a.iters_ran]), list(b.trace_step_size[:b.iters_ran]))
            fe = first_div([float(e) for e in a.trace.objective_evals[:a.iters_ran]], [float(e) for e in b.trace_objective_evals[:b.iters_ran]], 0.0)
            dx = float(np.max(np.abs(np.asarray(a.minimizer) - np.asarray(b.minimizer))))
            print(f"  centering {k} stage {j}: status {a.status}/{b.status} iters {a.iters_ran}/{b.iters_ran} first step-size divergence {fd} "
                  f"first evals divergence {fe} |dx|max {dx:.3e} f {a.objective:.17g}/{b.objective:.17g}")
    print(f"  centerings gpu {len(got_rets)} ref {len(ref.centering_results)}")


def main():
    ref = _oracle_run()
    cfg = cgo.setupCGConfig(1e-5, cgo.HagerZhang(), cgo.EnableTrace(), max_iters=1000)
    lsW = cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50)
    lsA = cgo.Backtracking(cgo.Armijo(1e-3), 0.9, 300, 50)
    cfgLS = cgo.setupCGConfig(1e-5, cgo.LiuStorrey(), cgo.EnableTrace(), max_iters=1000)
    cfgDFP = cgo.setupCGConfig(1e-5, cgo.setupBroydenFamily(1.0, 2), cgo.EnableTrace(), max_iters=1000)
    got = cgo.primalbarriermethod(cgo.BoxConstraints(-10.0, 10.0), "ObjBooth", X0, cfg, lsW,
                                  cgo.setupPrimalBarrierConfig(1e-8, 10.0, 100), (cfgDFP, lsA), (cfgLS, lsW))
    print("device objective: status", got.status, ref.status, "iters", got.iters_ran, ref.iters_ran, "t", got.t_final, ref.t_final)
    report("device barrier objective (device log)", got.centering_results, ref)
    # closure path: the same host loop, objective = the numpy evalbarrier (libm log) called back from the GPU engine
    con = N.CvxInequalityConstraint(4, 2)
    hdh = N.make_boxhdh([-10.0, -10.0], [10.0, 10.0])
    g0 = np.empty(2)
    t = N.booth(g0, np.array(X0)) * 10.0
    rets = []
    for i in range(len(ref.centering_results)):
        tt = t

        def fdf(g, x, tt=tt):
            return N.evalbarrier(con, g, N.booth, hdh, x, tt)
        rets.append(cgo.minimizeobjectivererun(fdf, np.array(X0), cfg, lsW, (cfgDFP, lsA), (cfgLS, lsW)))
        if rets[-1][-1].status != "success":
            break
        t *= 10.0
    report("closure path (numpy evalbarrier through the C callback)", rets, ref)


if __name__ == "__main__":
    main()
