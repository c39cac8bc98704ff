"""First-contact diagnostic on a real MI355X: parity vs oracle + kernel micro-benchmarks."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cgo_amd as cgo
from oracle import oracle as O

def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))

def compare(name, obj_dev, obj_orc, x0, beta_dev, beta_orc, ls_dev, ls_orc, eps=1e-5, max_iters=50):
    cfg = cgo.setupCGConfig(eps, beta_dev, cgo.EnableTrace(), max_iters=max_iters)
    s = cgo.Solver(obj_dev, cfg, ls_dev); s.enable_trial_log(); s.set_x0(x0); s.start()
    while not s.iterate(1 << 30): pass
    r = s.results(); la, lp, ld = s.trial_log(); s.close()
    ro = O.minimizeobjective(obj_orc, x0, O.cg_config(eps, beta_orc, max_iters), ls_orc, log_cap=100000)
    same_seq = len(la) == len(ro.log_a) and np.array_equal(la, ro.log_a)
    ndiv = next((i for i in range(min(len(la), len(ro.log_a))) if la[i] != ro.log_a[i]), None)
    print(f"{name}: status {r.status}/{ro.status} iters {r.iters_ran}/{ro.iters_ran} "
          f"f {r.objective:.15e}/{ro.objective:.15e} relx {rel(r.minimizer, ro.minimizer):.2e} "
          f"relg {rel(r.gradient, ro.gradient):.2e} trials {len(la)}/{len(ro.log_a)} same_steps {same_seq} first_div {ndiv} "
          f"launches {r.total_launches}")

ctx = cgo.default_context()
# Booth (examples/min.jl)
compare("booth/HZ", cgo.Booth(), O.objective("booth"), np.array([0.43, 1.23]), cgo.HagerZhang(), O.beta_config("HagerZhang"),
        cgo.setupStrongWolfeBisection(1e-5, 0.8), O.strong_wolfe(1e-5, 0.8), max_iters=1000)
for n in (2, 3, 31, 64, 1000, 100003):
    D = O.fill_uniform(n, 24, 1.0, 1000.0)
    x0 = np.ones(n)
    for bname, bd in (("PolakRibiere", cgo.PolakRibiere()), ("HagerZhang", cgo.HagerZhang()), ("DaiYuan", cgo.DaiYuan()),
                      ("LiuStorrey", cgo.LiuStorrey()), ("SallehAlhawarat", cgo.SallehAlhawarat()), ("YuanWangSheng", cgo.YuanWangSheng(0.1)),
                      ("HestenesStiefel", cgo.HestenesStiefel())):
        compare(f"quad n={n} {bname} SW", cgo.QuadDiag(D), O.objective("quad_diag", D=D), x0, bd,
                O.beta_config(bname), cgo.setupStrongWolfeBisection(1e-5, 0.8), O.strong_wolfe(1e-5, 0.8), eps=1e-9, max_iters=40)
    compare(f"quad n={n} HZ WolfeBis", cgo.QuadDiag(D), O.objective("quad_diag", D=D), x0, cgo.HagerZhang(),
            O.beta_config("HagerZhang"), cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50),
            O.wolfe_bisection("Wolfe", 1e-3, 0.9), eps=1e-9, max_iters=40)
    compare(f"quad n={n} PR YWL", cgo.QuadDiag(D), O.objective("quad_diag", D=D), x0, cgo.PolakRibiere(),
            O.beta_config("PolakRibiere"), cgo.WolfeBisection(cgo.YuanWeiLuWolfe(1e-3, 0.9, 1e-4), 100, 1e12, 50),
            O.wolfe_bisection("YuanWeiLuWolfe", 1e-3, 0.9, delta1=1e-4), eps=1e-9, max_iters=40)
for n in (2, 32, 1000):
    x0 = np.tile([-1.2, 1.0], n // 2) + 0.01 * O.fill_uniform(n, 7, -1, 1)
    compare(f"rosen n={n} HZ Wolfe", cgo.RosenbrockPaired(n), O.objective("rosenbrock_paired"), x0, cgo.HagerZhang(),
            O.beta_config("HagerZhang"), cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50),
            O.wolfe_bisection("Wolfe", 1e-3, 0.9), max_iters=60)
    compare(f"rosen n={n} PR SW", cgo.RosenbrockPaired(n), O.objective("rosenbrock_paired"), x0, cgo.PolakRibiere(),
            O.beta_config("PolakRibiere"), cgo.setupStrongWolfeBisection(1e-5, 0.8), O.strong_wolfe(1e-5, 0.8), max_iters=60)

# micro-benchmarks
for n in (10**6, 10**7, 10**8):
    obj = cgo.QuadDiagRandom(n, 24, 1.0, 1000.0)
    for kind, nm in ((2, "accept_dir_trial"), (1, "trial"), (3, "accept_dir"), (100, "dir"), (101, "beta_partials"), (4, "accept_only")):
        ms, b = cgo.bench_kernel(kind, n, 20, obj)
        print(f"bench n={n:.0e} {nm:18s} {ms*1e3:9.1f} us  {b/ms/1e6:8.1f} GB/s  ({b/ms/1e6/8000*100:.1f}% of 8 TB/s)")
    obj.close()
# solver throughput
for n, iters in ((10**6, 200), (10**7, 100), (10**8, 50)):
    obj = cgo.QuadDiagRandom(n, 24, 1.0, 1000.0)
    cfg = cgo.setupCGConfig(1e-300 if False else 1e-200, cgo.PolakRibiere(), cgo.EnableTrace(), max_iters=100000)
    s = cgo.Solver(obj, cfg, cgo.setupStrongWolfeBisection(1e-5, 0.1)); s.profile(True); s.set_x0_fill("constant", 1.0); s.start()
    s.iterate(10); s.profile_reset()
    t = time.time(); s.iterate(iters); dt = time.time() - t
    r = s.results(vectors=False)
    print(f"solve n={n:.0e}: {iters/dt:.1f} it/s, evals/iter {r.trace.objective_evals[-iters:].mean():.2f}, f={r.objective:.6e} status={r.status}")
    for k, v in s.profile_get().items():
        print(f"    {k:18s} launches {v['launches']:5d} avg {v['total_ms']/v['launches']*1e3:9.1f} us  {v['bytes_per_launch']/(v['total_ms']/v['launches'])/1e6:8.1f} GB/s")
    s.close(); obj.close()
