import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cgo_amd as cgo
for n, iters in ((10**4, 300), (10**6, 300), (10**7, 100)):
    obj = cgo.QuadDiagRandom(n, 24, 1.0, 1000.0)
    cfg = cgo.setupCGConfig(1e-200, cgo.PolakRibiere(), cgo.EnableTrace(), max_iters=100000)
    for prof in (False, True):
        s = cgo.Solver(obj, cfg, cgo.setupStrongWolfeBisection(1e-5, 0.1)); s.profile(prof); s.set_x0_fill("constant", 1.0); s.start()
        s.iterate(10); s.profile_reset()
        t = time.perf_counter(); s.iterate(iters); dt = time.perf_counter() - t
        r = s.results(vectors=False)
        L = r.total_launches
        print(f"publish={os.environ.get('CGO_HOST_PUBLISH','1')} prof={prof} n={n:.0e}: {iters/dt:9.1f} it/s  {dt/iters*1e6:7.1f} us/iter  evals/iter {r.trace.objective_evals[-iters:].mean():.2f}")
        if prof:
            for k, v in s.profile_get().items():
                print(f"      {k:18s} {v['launches']:5d} avg {v['total_ms']/v['launches']*1e3:8.1f} us")
        s.close()
    obj.close()
