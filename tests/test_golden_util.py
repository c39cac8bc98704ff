"""Loader for tests/golden/trajectories.json (generator: tests/golden/make_golden.py)."""
import json
import os

import numpy as np

from _cases import Case, O
from _suite import rosen_x0

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    with open(os.path.join(GOLD, "trajectories.json")) as f:
        data = json.load(f)
    out = []
    for item in data:
        d = dict(item["case"])
        n = d.pop("n")
        x0s, Ds = d.pop("x0"), d.pop("D")
        if x0s == ["ones"]:
            x0 = np.ones(n)
        elif x0s and x0s[0] == "rosen":
            x0 = rosen_x0(n, x0s[1], x0s[2])
        else:
            x0 = np.array(x0s, dtype=np.float64)
        D = O.fill_uniform(n, Ds[1], Ds[2], Ds[3]) if Ds else None
        c = Case(d.pop("name"), d.pop("objective"), n, x0, D=D, **d)
        e = dict(item["expect"])
        for k in ("minimizer", "trace_objective", "trace_grad_norm", "trace_step_size", "log_a"):
            e[k] = np.array(e[k], dtype=np.float64)
        e["trace_objective_evals"] = np.array(e["trace_objective_evals"], dtype=np.int64)
        out.append((c, e))
    return out
