import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from _cases import run_gpu, first_divergence
from _suite import parity_cases
name = sys.argv[1] if len(sys.argv) > 1 else "quad100003-SallehAlhawarat-SW"
c = [c for c in parity_cases(sizes=(1000, 100003)) if c.name == name][0]
for multi in ("0", "1000000000"):
    os.environ["CGO_MULTI_MIN_N"] = multi; os.environ["CGO_MULTI5_MIN_N"] = os.environ["CGO_MULTI7_MIN_N"] = "9000000000000000000"
    base = None
    for depth, chunk in (("0", 0), ("0", 3), ("1", 0), ("8", 0), ("32", 0), ("5", 3)):
        os.environ["CGO_CTL_DEPTH"] = depth
        r = run_gpu(c, chunk=chunk)
        if base is None:
            base = r
        print(f"multi={multi:>10s} depth={depth:>2s} chunk={chunk} iters={r.iters_ran} evals={r.total_fdf_evals} launches={r.total_launches} "
              f"served={r.controller_launches} f={r.objective.hex()} div={first_divergence(r, base)} x_equal={np.array_equal(r.minimizer, base.minimizer)} "
              f"trace_equal={np.array_equal(r.trace_objective, base.trace_objective)}")
