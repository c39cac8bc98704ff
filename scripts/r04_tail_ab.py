"""A/B of two library builds through child processes: the same cases must come out BITWISE equal; wall-clock per iteration
(host-driven, events off).  python3 scripts/r04_tail_ab.py <name=path.so> <name=path.so>"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def child():
    import numpy as np
    import cgo_amd as cgo
    from _cases import Case, quad_D, run_gpu, _product_structs, gpu_objective
    from _suite import rosen_x0
    os.environ["CGO_RESIDENT"] = "0"
    out = {}
    cases = []
    for n in (3001, 70000, 100003, 262144, 1000001, 3000001):
        cases.append(Case(f"q{n}", "quad_diag", n, np.ones(n), beta="PolakRibiere", D=quad_D(n), eps=1e-200, max_iters=12, c2=0.1))
        cases.append(Case(f"r{n}", "rosenbrock_paired", n + (n & 1), rosen_x0(n + (n & 1), 0.05, 3), beta="HagerZhang", ls="WolfeBisection", c1=1e-3, c2=0.9,
                          ls_max_iters=100, eps=1e-200, max_iters=10))
    for c in cases:
        r = run_gpu(c)
        h = hashlib.sha256()
        for a in (r.minimizer, r.gradient, r.trace_objective, r.trace_grad_norm, r.trace_step_size, r.log_phi, r.log_dphi):
            h.update(np.ascontiguousarray(a).tobytes())
        out[c.name] = (h.hexdigest()[:16], r.iters_ran, r.total_launches)
    tim = {}
    for tag, mk, n, iters in (("quad n=1e4", "q", 10000, 400), ("quad n=1e5", "q", 100000, 400), ("quad n=1e6", "q", 1000000, 300), ("rosen n=1e6", "r", 1000000, 300),
                              ("rosen n=1e7 (c3)", "r", 10000000, 200), ("quad n=3e6", "q", 3000000, 200), ("quad n=1.25e7", "q", 12500000, 200)):
        if mk == "q":
            c = Case(tag, "quad_diag", n, np.ones(n), beta="PolakRibiere", D=quad_D(n), eps=1e-200, max_iters=iters + 20, c2=0.1)
        else:
            c = Case(tag, "rosenbrock_paired", n, rosen_x0(n, 0.05, 3), beta="HagerZhang", ls="WolfeBisection", c1=1e-3, c2=0.9, ls_max_iters=100, eps=1e-200, max_iters=iters + 20)
        _, _lib, cfg, ls = _product_structs(c)
        obj = gpu_objective(c)
        s = cgo.Solver(obj, cfg, ls, cgo.SolverPolicy(resident=False, controller_depth=0))
        s.set_x0(c.x0); s.start(); s.iterate(20)
        best, per = 1e9, iters // 4
        for w in range(4):
            t0 = time.perf_counter(); s.iterate(per); dt = time.perf_counter() - t0
            best = min(best, dt / per)
        tim[tag] = round(best * 1e6, 2)
        s.close(); obj.close()
    print("RESULT " + json.dumps({"hashes": out, "us_per_iteration": tim}))


def main():
    libs = dict(a.split("=", 1) for a in sys.argv[1:])
    res = {}
    for rep in (1, 2):
        for name, lib in libs.items():
            env = dict(os.environ, CGO_LIB_PATH=os.path.join(ROOT, lib))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=900)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            if not line:
                print(p.stdout[-2000:], p.stderr[-2000:]); sys.exit(1)
            res[(name, rep)] = json.loads(line[0][7:])
            print(name, rep, res[(name, rep)]["us_per_iteration"], flush=True)
    names = list(libs)
    a, b = res[(names[0], 1)]["hashes"], res[(names[1], 1)]["hashes"]
    diff = [k for k in a if a[k] != b[k]]
    print("cases:", len(a), "bitwise different between the builds:", diff)
    sys.exit(1 if diff else 0)


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
