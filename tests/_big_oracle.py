"""Child-process runner of the CPU oracle for the BASELINE-size parity tests (tests/test_baseline_sizes.py).

TEST INFRASTRUCTURE.  `python tests/_big_oracle.py <config> <n> <iters> <out.npz> [omp]`

Runs oracle/cgo_oracle.c on one BASELINE.json configuration at its full size and stores what
`_cases.assert_parity` compares.  With `omp` the -fopenmp build of the SAME source is loaded in this fresh process
(libgomp reads OMP_NUM_THREADS once, when it is first loaded — the parent sets it to the container's CPU share):
its reductions are `omp parallel for reduction(+)` over static chunks — another valid summation order of the same
restatement, like the numpy oracle's BLAS order; the two agree to 7e-14 with the 1-thread build at n = 1e8 over six
iterations (measured).  Never touches the GPU.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def baseline_case(config: str, n: int, iters: int):
    """The Case of one BASELINE.json config at size n (same definitions as bench.py's WORKLOADS / SURVEY.md §8(d))."""
    from _cases import Case, quad_D
    from oracle import oracle as O
    if config in ("c2", "c5"):   # separable quadratic, D_i = 1 + 999·U_i (seed 24), x0 = 1, PR + StrongWolfeBisection(1e-5, 0.1)
        return Case(f"{config}-n{n:.0e}", "quad_diag", n, np.ones(n), beta="PolakRibiere", c1=1e-5, c2=0.1, eps=1e-200,
                    max_iters=iters, D=quad_D(n))
    if config == "c3":           # extended Rosenbrock, HZ + WolfeBisection(Wolfe(1e-3, 0.9), 100, 1e12, 50)
        return Case(f"c3-n{n:.0e}", "rosenbrock_paired", n, np.tile([-1.2, 1.0], n // 2), beta="HagerZhang", ls="WolfeBisection",
                    cond="Wolfe", c1=1e-3, c2=0.9, ls_max_iters=100, max_step_size=1e12, feas_max_iters=50, eps=1e-200, max_iters=iters)
    if config == "c4":           # log-sum-exp + λ/2‖x‖², λ = 1e-2/n, x0 = 5(2U−1) (seed 24), L-BFGS m = 10, StrongWolfe(1e-5, 0.9)
        return Case(f"c4-n{n:.0e}", "lse", n, O.fill_uniform(n, 24, -5.0, 5.0), beta="LBFGS", m=10, c1=1e-5, c2=0.9, eps=1e-200,
                    max_iters=iters, lam=1e-2 / n)
    raise KeyError(config)


def main():
    config, n, iters, out = sys.argv[1], int(float(sys.argv[2])), int(sys.argv[3]), sys.argv[4]
    omp = len(sys.argv) > 5 and sys.argv[5] == "omp"
    from oracle import oracle as O
    if omp:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        O.use_openmp(True)
    from _cases import run_oracle
    r = run_oracle(baseline_case(config, n, iters))
    np.savez(out, objective=r.objective, minimizer=r.minimizer, gradient=r.gradient, iters_ran=r.iters_ran, status=r.status,
             trace_objective=r.trace_objective, trace_grad_norm=r.trace_grad_norm, trace_step_size=r.trace_step_size,
             trace_objective_evals=r.trace_objective_evals, log_a=r.log_a, log_phi=r.log_phi, log_dphi=r.log_dphi,
             total_fdf_evals=r.total_fdf_evals)


if __name__ == "__main__":
    main()
