#!/usr/bin/env python3
"""Soak of the resident solver under GPU SHARING: P processes on one GPU, each running multi-workgroup resident solves
(245 workgroups per launch: they wait for one another's rows) at the same time.  Two persistent launches from different
processes can each be partially resident and starve each other; the bounded polls then give a slice up, the
launch-per-trial engine redoes it, and the solver leaves the resident path — every solve must still end with the
results of an undisturbed run, bit for bit in the step sequence and ≤ 1e-10 on the iterate.

    python3 scripts/soak_resident.py [processes = 3] [solves per process = 40] [n = 1000000]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, ROOT)
import cgo_amd as cgo
ctx = cgo.Context(0)
n, solves, rank = N, SOLVES, RANK
obj = cgo.QuadDiagRandom(n, 24 + 7 * rank, 1.0, 1000.0, ctx)
cfg = cgo.setupCGConfig(1e-200, cgo.PolakRibiere(), cgo.EnableTrace(), max_iters=60)
ls = cgo.setupStrongWolfeBisection(1e-5, 0.1)
def solve(resident):
    os.environ["CGO_RESIDENT"] = "1" if resident else "0"
    s = cgo.Solver(obj, cfg, ls)
    s.set_x0_fill("constant", 1.0); s.start()
    while not s.iterate(20):
        pass
    r = s.results()
    st = s.resident_stats(); gave = s.resident_gave_up
    s.close()
    return r, st, gave
ref, _, _ = solve(False)
gave_up = slices = iters = 0
t0 = time.time()
worst = 0.0
for k in range(solves):
    r, st, g = solve(True)
    gave_up += g; slices += st[0]; iters += st[1]
    assert r.status == ref.status and r.iters_ran == ref.iters_ran, (r.status, ref.status)
    assert np.array_equal(r.trace.step_size, ref.trace.step_size) and np.array_equal(r.trace.objective_evals, ref.trace.objective_evals)
    d = float(np.linalg.norm(r.minimizer - ref.minimizer) / np.linalg.norm(ref.minimizer))
    worst = max(worst, d)
    assert d <= 1e-10, d
print(json.dumps(dict(rank=rank, solves=solves, resident_slices=slices, resident_iterations=iters, slices_given_up=gave_up,
                      worst_rel_diff_vs_host_driven=worst, seconds=round(time.time() - t0, 2))), flush=True)
"""


def main():
    procs_n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    solves = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1000000
    procs = []
    for r in range(procs_n):
        code = f"ROOT={ROOT!r}; N={n}; SOLVES={solves}; RANK={r}\n" + WORKER
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    ok = True
    out = []
    for p in procs:
        o = p.communicate(timeout=900)[0]
        line = [l for l in o.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            ok = False
            print(o[-2000:])
        else:
            out.append(json.loads(line[-1]))
            print(line[-1])
    print(json.dumps(dict(processes=procs_n, all_ok=ok, slices_given_up=sum(d["slices_given_up"] for d in out),
                          resident_iterations=sum(d["resident_iterations"] for d in out))))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
