"""Generates tests/golden/*.json from the numpy restatement (oracle/cgo_oracle_np.py).

The reference is Julia and cannot run here (no julia binary, no network), and its
own tests hold no solver vectors (test/runtests.jl:7-44 pins only the Booth
gradient), so these fixtures are outputs of OUR restatement of the reference's
text, not of the reference itself: "parity unpinned" beyond the hand-derived
KATs in kat.json and the Booth known answer.

    python tests/golden/make_golden.py     # rewrites the fixtures
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _cases import run_numpy  # noqa: E402
from _suite import parity_cases  # noqa: E402


def case_to_json(c):
    d = {k: getattr(c, k) for k in ("name", "objective", "n", "beta", "mu", "m", "ls", "c1", "c2", "growth",
                                     "ls_max_iters", "zoom_max_iters", "cond", "delta1", "max_step_size",
                                     "feas_max_iters", "discount", "eps", "max_iters", "lam")}
    # inputs are regenerated from the counter-based RNG (oracle.fill_uniform == device k_fill)
    if c.objective == "quad_diag":
        d["x0"], d["D"] = ["ones"], ["uniform", 24, 1.0, 1000.0]
    elif c.objective == "rosenbrock_paired":
        d["x0"], d["D"] = ["rosen", 0.01, 7], None
    else:
        d["x0"], d["D"] = [float(v) for v in c.x0], None
    return d


def main():
    cases = [c for c in parity_cases(sizes=(31, 64, 1000), small_only=True)]
    out = []
    for c in cases:
        r = run_numpy(c)
        out.append(dict(case=case_to_json(c), expect=dict(
            status=r.status, iters_ran=int(r.iters_ran), objective=float(r.objective),
            minimizer=[float(v) for v in r.minimizer], gradient_norm=float(np.linalg.norm(r.gradient)),
            trace_objective=[float(v) for v in r.trace_objective],
            trace_grad_norm=[float(v) for v in r.trace_grad_norm],
            trace_step_size=[float(v) for v in r.trace_step_size],
            trace_objective_evals=[int(v) for v in r.trace_objective_evals],
            log_a=[float(v) for v in r.log_a])))
    with open(os.path.join(HERE, "trajectories.json"), "w") as f:
        json.dump(out, f)
    # hand-derived known answers (exact rationals)
    kat = dict(
        beta=dict(g_next=[1.0, 2.0], g=[3.0, -1.0], u=[-3.0, 1.0],
                  expect={"HagerZhang": 62 / 81, "YuanWangSheng": 62 / 81, "SallehAlhawarat": 4 / 9,
                          "LiuStorrey": -4 / 9, "HestenesStiefel": 4 / 9, "PolakRibiere": 2 / 5,
                          "DaiYuan": 5 / 9},
                  partials=dict(gtu=-1.0, gtgt=5.0, gtg=1.0, yy=13.0, uy=9.0, ygt=4.0, gg=10.0, gu=-10.0, uu=10.0),
                  updatedir_HZ=dict(u_new=[-89 / 27, -100 / 81], gu=-467 / 81)),
        booth_first_iteration=dict(x0=[0.43, 1.23], f0=25.3602, g0=[-19.86, -22.26], dphi0=-889.9272,
                                   trial_a=[1.0, 0.5, 0.25, 0.125, 0.0625],
                                   trial_phi=[7121.7378, 1576.9728, 302.02245, 38.9053125, 0.936253125],
                                   dphi_accept=108.3609, a_star=0.0625, evals=5),
        booth_minimizer=[1.0, 3.0])
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("wrote", len(out), "trajectories")


if __name__ == "__main__":
    main()
