"""Long-horizon parity against the ARBITER (oracle/cgo_oracle.c built with -DORC_EXACT_SUMS).  TEST INFRASTRUCTURE.

The 1e-10 bar of `_cases.assert_parity` holds "for the same step sequence" over 5–12 iterations; `bench.py` times
100–250.  Over such horizons no two valid implementations of the reference stay on one trajectory for ever: the order of
a reduction is unspecified (`LinearAlgebra.dot` → BLAS), every order perturbs the sums in their last bits, and nonlinear
CG amplifies that (Hager–Zhang on Rosenbrock: ¼ digit per iteration).  So the long-horizon statement is made against
the trajectory the reference's ALGORITHM defines when its reductions are (almost) exact — the arbiter — and has three parts:

  1. BRANCHES.  Walk the evaluation logs (every evalϕdϕ! of every line search) of an implementation and of the arbiter
     together.  Up to the first step that differs bit for bit they took the same decisions.  Where they part, the
     arbiter's own decision margin on the PREVIOUS evaluation (the one whose branch chose that step: `log_margin`,
     |lhs − rhs| / scale of the inequality, cgo_oracle.c `note_margin`) must be small against the drift measured so far:
     the arbiter's branch is taken wherever its margin exceeds max(1e-9, 100 × drift).
  2. CURVE.  Per outer iteration k up to that point: |f_k − f_k^arb| / |f_k^arb| and the same for ‖g_k‖ — the measured
     error-vs-exact curve of the implementation.
  3. ITERATES.  At the checkpoints up to that point: ‖x_k − x_k^arb‖ / ‖x_k^arb‖ of the implementation under test must
     not exceed K × that of the double-precision oracle (another valid summation order) + a floor.

`python tests/_long_horizon.py <config> <n> <iters> <out.npz> <exact|omp|c> <ckpt,ckpt,…>` runs one oracle build in a
fresh process (OMP_NUM_THREADS is read once per process) and stores logs, traces, margins and checkpoint iterates.
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


@dataclass
class Traj:
    """What one implementation leaves behind for the comparison."""
    name: str
    log_a: np.ndarray            # step of every evaluation
    evals: np.ndarray            # evaluations per accepted iteration (trace.objective_evals)
    f: np.ndarray                # trace.objective
    gnorm: np.ndarray            # trace.grad_norm
    step: np.ndarray             # trace.step_size
    status: str
    iters_ran: int
    snap_iters: np.ndarray       # checkpoints reached
    snap_x: np.ndarray           # [len(snap_iters), n]
    log_margin: np.ndarray | None = None


def run_oracle_build(config: str, n: int, iters: int, build: str, ckpts) -> Traj:
    from _big_oracle import baseline_case
    from _cases import _orc_ls
    from oracle import oracle as O
    if build == "exact":
        O.use_exact(True)
    elif build == "omp":
        O.use_openmp(True)
    c = baseline_case(config, n, iters)
    obj = O.objective(c.objective, D=c.D, lam=c.lam)
    cfg = O.cg_config(c.eps, O.beta_config(c.beta, c.mu, c.m), c.max_iters, True)
    r = O.minimizeobjective(obj, c.x0, cfg, _orc_ls(c), log_cap=64 * iters + 1024, snap_iters=list(ckpts))
    assert bool(O.lib().orc_exact_sums()) == (build == "exact")
    return Traj(build, r.log_a, r.trace_objective_evals, r.trace_objective, r.trace_grad_norm, r.trace_step_size, r.status,
                r.iters_ran, r.snap_iters, r.snap_x, r.log_margin)


def save(t: Traj, path: str):
    np.savez(path, name=t.name, log_a=t.log_a, evals=t.evals, f=t.f, gnorm=t.gnorm, step=t.step, status=t.status,
             iters_ran=t.iters_ran, snap_iters=t.snap_iters, snap_x=t.snap_x,
             log_margin=t.log_margin if t.log_margin is not None else np.zeros(0))


def load(path: str) -> Traj:
    d = np.load(path)
    lm = d["log_margin"]
    return Traj(str(d["name"]), d["log_a"], d["evals"], d["f"], d["gnorm"], d["step"], str(d["status"]), int(d["iters_ran"]),
                d["snap_iters"], d["snap_x"], lm if lm.size else None)


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def compare(t: Traj, arb: Traj) -> dict:
    """→ where `t` leaves the arbiter's step sequence, the arbiter's margin there, the error curves up to there."""
    m = min(len(t.log_a), len(arb.log_a))
    same = t.log_a[:m].view(np.uint64) == arb.log_a[:m].view(np.uint64)
    first = int(np.argmin(same)) if not same.all() else m      # index of the first evaluation whose step differs
    parted = first < max(len(t.log_a), len(arb.log_a)) and not (first == m and len(t.log_a) == len(arb.log_a))
    # iterations completed identically: those whose evaluations all lie before `first`
    ce = np.cumsum(arb.evals)
    k_same = int(np.searchsorted(ce, first, side="right")) if parted else min(len(t.f), len(arb.f))
    k_same = min(k_same, len(t.f), len(arb.f))
    # the decision that chose evaluation `first` was taken on evaluation first − 1 (or is the accepted step of the previous
    # line search, decided there as well)
    margin = float(arb.log_margin[first - 1]) if (parted and arb.log_margin is not None and first >= 1) else None
    f_err = np.abs(t.f[:k_same] - arb.f[:k_same]) / np.maximum(np.abs(arb.f[:k_same]), 1e-300)
    g_err = np.abs(t.gnorm[:k_same] - arb.gnorm[:k_same]) / np.maximum(np.abs(arb.gnorm[:k_same]), 1e-300)
    x_err = {}
    for j, k in enumerate(arb.snap_iters):
        if k <= k_same and j < len(t.snap_iters) and t.snap_iters[j] == k:
            x_err[int(k)] = rel(t.snap_x[j], arb.snap_x[j])
    return dict(name=t.name, evaluations_in_common=first, iterations_in_common=k_same, parted=bool(parted),
                arbiter_margin_where_parted=margin, f_err=f_err, g_err=g_err, x_err=x_err,
                min_margin_before=(float(arb.log_margin[:max(first - 1, 0)].min()) if arb.log_margin is not None and first > 1 else None))


def drift_before(cmp_: dict) -> float:
    """The largest relative error seen on f, ‖g‖ and the checkpoint iterates before the two trajectories part."""
    d = 0.0
    if len(cmp_["f_err"]):
        d = max(d, float(cmp_["f_err"].max()), float(cmp_["g_err"].max()))
    if cmp_["x_err"]:
        d = max(d, max(cmp_["x_err"].values()))
    return d


def summary(cmp_: dict, every: int = 10) -> dict:
    """JSON-able digest: the curve sampled every `every` iterations."""
    ks = [k for k in range(every, cmp_["iterations_in_common"] + 1, every)]
    return dict(name=cmp_["name"], evaluations_in_common=cmp_["evaluations_in_common"], iterations_in_common=cmp_["iterations_in_common"],
                parted=cmp_["parted"], arbiter_margin_where_parted=cmp_["arbiter_margin_where_parted"],
                min_arbiter_margin_before=cmp_["min_margin_before"],
                f_rel_err={str(k): float(cmp_["f_err"][k - 1]) for k in ks}, gnorm_rel_err={str(k): float(cmp_["g_err"][k - 1]) for k in ks},
                x_rel_err={str(k): v for k, v in sorted(cmp_["x_err"].items())})


def main():
    config, n, iters, out, build = sys.argv[1], int(float(sys.argv[2])), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    ckpts = [int(v) for v in sys.argv[6].split(",")] if len(sys.argv) > 6 and sys.argv[6] else []
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    save(run_oracle_build(config, n, iters, build, ckpts), out)


if __name__ == "__main__":
    main()
