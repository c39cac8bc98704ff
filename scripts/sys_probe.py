"""solvesystem on the GPU: iterations/s and per-kernel rates at a few sizes (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cgo_amd as cgo

for n in (10**6, 10**7, 10**8):
    obj = cgo.QuadDiagRandom(n, 24, 1.0, 2.0)
    cfg = cgo.setupCGConfig(1e-200, cgo.HagerZhang(), cgo.EnableTrace(), max_iters=1000)
    s = cgo.Solver(obj, cfg, cgo.setupLinesearchSolveSys(0.5))
    s.set_x0_fill("constant", 1.0)
    s.start()
    s.iterate(3)
    s.profile(True); s.profile_reset()
    t0 = time.perf_counter()
    fin = s.iterate(20)
    r = s.results(vectors=False)
    dt = time.perf_counter() - t0
    prof = s.profile_get()
    k = r.trace.objective_evals[3:r.iters_ran]
    print(f"n={n:.0e} {20/dt:8.1f} it/s  status={r.status} iters={r.iters_ran} trials/iter={float((k+1).mean()):.1f} family={s.kernel_family()}")
    for name, v in prof.items():
        avg = v['total_ms'] / v['launches']
        print(f"      {name:18s} {v['launches']:5d} launches avg {avg*1e3:9.1f} us {v['bytes_per_launch']/avg/1e6:8.0f} GB/s")
    s.close(); obj.close()
