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
from _suite import parity_cases, sys_cases  # noqa: E402


def case_to_json(c):
    d = {k: getattr(c, k) for k in ("name", "objective", "n", "beta", "mu", "m", "ls", "c1", "c2", "growth",
                                     "ls_max_iters", "zoom_max_iters", "cond", "delta1", "max_step_size",
                                     "feas_max_iters", "discount", "eps", "max_iters", "lam",
                                     "sys_s", "sys_sigma", "sys_rho", "sys_max_iters")}
    # inputs are regenerated from the counter-based RNG (oracle.fill_uniform == device k_fill)
    if c.objective == "quad_diag":
        d["x0"], d["D"] = ["ones"], ["uniform", 24, float(c.extra.get("D_lo", 1.0)), float(c.extra.get("D_hi", 1000.0))]
    elif c.objective == "rosenbrock_paired":
        d["x0"], d["D"] = c.extra.get("x0", ["rosen", 0.01, 7]), None
    else:
        d["x0"], d["D"] = [float(v) for v in c.x0], None
    return d


def main():
    cases = [c for c in parity_cases(sizes=(31, 64, 1000), small_only=True)]
    cases += [c for c in sys_cases(sizes=(31, 64, 1000), small_only=True)]   # solvesystem (solve_system.jl)
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
        booth_minimizer=[1.0, 3.0],
        # solvesystem on the identity system g(x) = x (quad_diag, D = 1), x0 = (3, 4), HagerZhang,
        # LinesearchSolveSys(ρ = 0.5, σ = 0.5, s = 0.25); by hand from solve_system.jl, all values exact:
        #  it 1: u0 = −x0, ‖u0‖² = 25.  a = 0.25: z = 0.75·x0 = (2.25, 3), ‖g(z)‖ = 3.75, −dϕ = 18.75,
        #        σ·a·‖g(z)‖·‖u‖² = 11.71875 → accepted at index 0 (:50-53).  m = a·dϕ/‖g(z)‖² = −1/3 (:248),
        #        x_next = x0 + m·g(z) = (2.25, 3) (:250), f = ½‖x‖² = 7.03125, ‖g‖ = 3.75.
        #        getβ(HZ): y = (−0.75, −1), R = u·y = 6.25, y·y = 1.5625, y·g⁺ = −4.6875, u·g⁺ = −18.75 →
        #        β = (−4.6875 + 9.375)/6.25 = 0.75;  u1 = −g⁺ + β·u0 = (−4.5, −6), ‖u1‖² = 56.25.
        #  it 2: a = 0.25: z = x1 + a·u1 = (1.125, 1.5), ‖g(z)‖ = 1.875, dϕ = −14.0625, rhs = 13.18359375 →
        #        accepted at index 0; m = 0.25·(−14.0625)/3.515625 = −1.  The reference adds m·g(z) to the
        #        x_next BUFFER, which still holds x0 (:172-178 — never re-based on x, swapped at :194):
        #        x2 = (3, 4) − (1.125, 1.5) = (1.875, 2.5), whereas re-basing would give (1.125, 1.5).
        solvesystem_first_iterations=dict(
            x0=[3.0, 4.0], D=[1.0, 1.0], s=0.25, sigma=0.5, rho=0.5,
            x1=[2.25, 3.0], f1=7.03125, norm_g1=3.75, a1=0.25, evals_index1=0, beta1=0.75, u1=[-4.5, -6.0],
            a2=0.25, dphi_second_search_first_trial=-14.0625, trial_steps=[0.25, 0.25],
            x2_reference=[1.875, 2.5], x2_if_rebased=[1.125, 1.5]))
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("wrote", len(out), "trajectories")


if __name__ == "__main__":
    main()
