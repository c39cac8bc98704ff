#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE outer iteration of minimizeobjective (reference src/engine/optim.jl:50-160:
line search + getβ + iterate/direction update) on BASELINE config 5: separable quadratic
f = ½ Σ D_i x_i², D_i = 1 + 999·U_i (counter RNG, seed 24), n = 1e8, x0 = 1, Polak–Ribière β,
StrongWolfeBisection(c1 = 1e-5, c2 = 0.1).  With N > 1 the SAME n = 1e8 state vector is
sharded contiguously over the N GPUs (strong scaling); every fused launch ends in one exchange of
its scalar block (≤ 56 doubles per rank): through a host shared-memory mailbox the finalize kernels
publish into (default on one node), the library's RCCL all-gather over xGMI, or a torch.distributed
callback — whichever passes a sharded self-test first (--comm).  Inputs are generated on the device
and are resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0 (contract + `roofline` + `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask ∩ cgroup CPU quota (a GPU box hands a
    container a share of its cores; os.cpu_count() reports the whole machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_baseline(n_sample: int, n_full: int, c1: float, c2: float, all_cores: bool = False):
    """Times the oracle (C restatement of the reference's pass structure; 1 thread, or the -fopenmp
    build on every host core) on a bounded sample of the same workload; iterations/s scaled to n_full."""
    import numpy as np
    from oracle import oracle as O
    if all_cores and "OMP_NUM_THREADS" not in os.environ:
        os.environ["OMP_NUM_THREADS"] = str(usable_cores())   # read by libgomp when the OpenMP build is loaded
    if all_cores:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # idle team members must not burn the container's CPU quota
    O.use_openmp(all_cores)
    D = O.fill_uniform(n_sample, 24, 1.0, 1000.0)
    x0 = np.ones(n_sample)
    obj = O.objective("quad_diag", D=D)
    ls = O.strong_wolfe(c1, c2)

    def run(iters):
        t = time.perf_counter()
        r = O.minimizeobjective(obj, x0, O.cg_config(1e-200, O.beta_config("PolakRibiere"), iters, True), ls)
        return time.perf_counter() - t, r
    w, k = 4, 16
    run(1)                      # page in the buffers, spin up the OpenMP team
    t_w, _ = run(w)
    t_k, r = run(w + k)
    dt = t_k - t_w              # iterations w+1..w+k
    if dt < 0.25 * t_k * k / (w + k):   # timer noise on a tiny sample: fall back to the whole run
        dt = t_k * k / (w + k)
    its = k / dt
    evals = float(r.trace_objective_evals[w:].mean())
    O.use_openmp(False)
    cores = (int(os.environ.get("OMP_NUM_THREADS", 0)) or usable_cores()) if all_cores else 1
    return dict(value=its * (n_sample / n_full), unit="iterations/s", cores=cores, kind="port",
                sample=(f"oracle/cgo_oracle.c (faithful pass structure, {cores} thread{'s' if cores > 1 else ''}) on the first n={n_sample:.0e} "
                        f"elements of the same quadratic, outer iterations {w + 1}..{w + k} "
                        f"({evals:.2f} trials/iter, {its:.2f} it/s at n={n_sample:.0e}), scaled by n ratio to n={n_full:.0e}"),
                host_cores_available=usable_cores(), host_cores_machine=os.cpu_count())


def cpu_baseline_child(n_sample: int, n_full: int, c1: float, c2: float, all_cores: bool):
    """Runs cpu_baseline in a fresh child process: libgomp reads OMP_NUM_THREADS once, when it is first
    loaded — in this process torch has loaded it long before, and its default (every core of the machine,
    256 on the GPU boxes) oversubscribes the container's CPU quota 16-fold (measured: 0.6–0.9 it/s against
    14.9 with 16 threads).  The child never touches the GPU."""
    import subprocess
    env = dict(os.environ)
    if all_cores:
        env["OMP_NUM_THREADS"] = str(usable_cores())
    code = ("import json, sys; sys.path.insert(0, %r); import bench; "
            "print(json.dumps(bench.cpu_baseline(%d, %d, %r, %r, all_cores=%r)))" % (ROOT, n_sample, n_full, c1, c2, all_cores))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("cpu_baseline child failed: " + r.stderr[-400:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", dest="n", type=float, default=None, help="global problem size (default: the workload's)")
    ap.add_argument("--workload", default="c5", choices=["c5", "c2", "c3", "c4"],
                    help="BASELINE.json config: c5 (default, headline) quadratic n=1e8 PR-CG; c2 quadratic n=1e6 PR-CG; "
                         "c3 extended Rosenbrock n=1e7 HZ + WolfeBisection; c4 log-sum-exp n=1e7 L-BFGS m=10")
    ap.add_argument("--beta", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comm", default="auto", choices=["auto", "shm", "rccl", "torch"],
                    help="scalar exchange: auto = host shared-memory mailbox (lowest latency, one node), else the library's "
                         "RCCL communicator, else a torch.distributed callback")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for rendezvous/barriers (gloo: rehearsal with several ranks on one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch  # device memory plumbing / torch.distributed only; loads the HIP runtime first
    import torch.distributed as dist
    import numpy as np
    import cgo_amd as cgo

    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    on_gpu = args.backend == "nccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")

    ctx = cgo.Context(dev_index)
    comm_used = "none"
    if world > 1:
        def torch_allgather(send):
            t = torch.from_numpy(send.copy())
            if on_gpu:
                t = t.cuda()
            out = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t)
            return out.cpu().numpy()
        def agree(ok: int) -> bool:
            flag = torch.tensor([ok], device="cuda" if on_gpu else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        def fresh_ctx():
            nonlocal ctx
            ctx.close()
            ctx = cgo.Context(dev_index)

        def selftest() -> bool:
            """A tiny sharded solve: every rank must finish it and hold the same objective."""
            try:
                o = cgo.QuadDiagRandom(8192, 24, 1.0, 1000.0, ctx)
                s0 = cgo.Solver(o, cgo.setupCGConfig(1e-200, cgo.DaiYuan(), cgo.EnableTrace(), max_iters=4),
                                cgo.setupStrongWolfeBisection(1e-5, 0.1))
                s0.set_x0_fill("constant", 1.0); s0.start(); s0.iterate(4)
                f = s0.results(vectors=False).objective
                s0.close(); o.close()
                t = torch.tensor([f, -f], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t[0].item()) == f and float(t[1].item()) == -f
            except Exception as e:
                print(f"[rank {rank}] exchange self-test failed: {e}", file=sys.stderr)
                return False

        comm_used = None
        if args.comm in ("auto", "shm"):
            ok = 1
            try:
                box = [f"/cgo_bench_{os.getpid()}_{int(time.time() * 1e6) & 0xFFFFFF}" if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                if rank == 0:
                    ctx.set_comm_shm(rank, world, box[0], True)
                dist.barrier()
                if rank != 0:
                    ctx.set_comm_shm(rank, world, box[0], False)
                dist.barrier()
                if rank == 0:
                    cgo.shm_unlink(box[0])
            except Exception as e:
                print(f"[rank {rank}] shared-memory mailbox unavailable ({e})", file=sys.stderr)
                ok = 0
            if agree(ok) and agree(int(selftest())):
                comm_used = "host shared-memory mailbox (finalize kernels publish into a POSIX shm segment)"
            else:
                fresh_ctx()
        if comm_used is None and args.comm in ("auto", "rccl") and on_gpu:
            ok = 1
            try:
                box = [cgo.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                ctx.set_comm_rccl(rank, world, box[0])
            except Exception as e:  # fall back together, never silently
                print(f"[rank {rank}] RCCL communicator failed ({e})", file=sys.stderr)
                ok = 0
            if agree(ok) and agree(int(selftest())):
                comm_used = "rccl all-gather (library communicator)"
            else:
                fresh_ctx()
        if comm_used is None:
            ctx.set_comm_callback(rank, world, torch_allgather)
            comm_used = "torch.distributed callback"

    W = {  # workload → (default n, default β, description)
        "c5": (1e8, "PolakRibiere", "separable quadratic f=1/2 sum D_i x_i^2, D_i=1+999*U_i (splitmix64 counter RNG, seed 24), x0=1, "
                                    "StrongWolfeBisection(c1=1e-5,c2=0.1,growth=2) [BASELINE config 5]"),
        "c2": (1e6, "PolakRibiere", "separable quadratic f=1/2 sum D_i x_i^2, D_i=1+999*U_i (seed 24), x0=1, "
                                    "StrongWolfeBisection(c1=1e-5,c2=0.1,growth=2) [BASELINE config 2]"),
        "c3": (1e7, "HagerZhang", "extended (paired) Rosenbrock, x0=(-1.2,1,...), WolfeBisection(Wolfe(1e-3,0.9),100,1e12,50) "
                                  "[BASELINE config 3]"),
        "c4": (1e7, "LBFGS", "log-sum-exp f=log sum exp(x_i)+lambda/2|x|^2, lambda=1e-2/n, x0_i=5(2U_i-1) (seed 24), L-BFGS m=10, "
                             "StrongWolfeBisection(c1=1e-5,c2=0.9) [BASELINE config 4]"),
    }[args.workload]
    n = int(args.n if args.n else W[0])
    bname = args.beta or W[1]
    c1, c2 = 1e-5, 0.1
    beta = {"PolakRibiere": cgo.PolakRibiere(), "HagerZhang": cgo.HagerZhang(), "DaiYuan": cgo.DaiYuan(),
            "LBFGS": cgo.LBFGS(10)}[bname]
    cfg = cgo.setupCGConfig(1e-200, beta, cgo.EnableTrace(), max_iters=args.warmup + args.steps + 8)
    if args.workload in ("c5", "c2"):
        obj = cgo.QuadDiagRandom(n, 24, 1.0, 1000.0, ctx)
        s = cgo.Solver(obj, cfg, cgo.setupStrongWolfeBisection(c1, c2))
        s.set_x0_fill("constant", 1.0)
    elif args.workload == "c3":
        obj = cgo.RosenbrockPaired(n, ctx)
        s = cgo.Solver(obj, cfg, cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50))
        s.set_x0_fill("alternate", -1.2, 1.0)
    else:
        obj = cgo.LogSumExp(n, 1e-2 / n, ctx)
        s = cgo.Solver(obj, cfg, cgo.setupStrongWolfeBisection(1e-5, 0.9))
        s.set_x0_fill("uniform", -5.0, 5.0, seed=24)
    s.start()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        s.iterate(args.warmup)
    no_prof = os.environ.get("CGO_BENCH_NO_PROFILE") == "1"   # experiments only: timing without the HIP-event ring
    s.profile(not no_prof)
    s.profile_reset()
    ctl0 = s.controller_launches()
    barrier()
    t0 = time.perf_counter()
    finished = s.iterate(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = s.profile_get()
    r = s.results(vectors=False)
    steps_done = r.iters_ran - args.warmup
    if finished or steps_done != args.steps:
        raise SystemExit(f"solver stopped early: status={r.status} after {r.iters_ran} iterations")

    if rank == 0 and no_prof:
        print(json.dumps({"value": args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "profile": "off",
                          "controller_armed_launches_per_iteration": (s.controller_launches() - ctl0) / args.steps}))
    elif rank == 0:
        trials = float(r.trace.objective_evals[args.warmup:].mean())
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        kname, kv = dom
        avg_ms = kv["total_ms"] / kv["launches"]
        achieved = kv["bytes_per_launch"] / avg_ms / 1e6  # GB/s
        total_alg_bytes = sum(v["bytes_per_launch"] * v["launches"] for v in prof.values())
        kernel_ms = sum(v["total_ms"] for v in prof.values())
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if world == 1 and n == 10**8 and args.workload == "c5" and os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(kname, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": ("CG iterations/sec at n=1e8 (outer iterations of minimizeobjective, PR-CG)" if args.workload == "c5" and n == 10**8
                       else f"outer iterations/sec of minimizeobjective, workload {args.workload}, n={n:.0e}, {bname}"),
            "value": args.steps / dt,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"n={n:.0e}, {bname}: " + W[2],
                "n": n,
                "n_per_gpu": obj.n_local,
                "sharding": "contiguous n/N per GPU; one exchange of 10–56 doubles per fused launch" if world > 1 else "single GPU",
                "comm": comm_used,
                "trials_per_iteration": trials,
                "launches_per_iteration": sum(v["launches"] for v in prof.values()) / args.steps,
                "controller_armed_launches_per_iteration": (s.controller_launches() - ctl0) / args.steps,
            },
            "achieved_hbm_gbps_per_gpu_all_kernels": total_alg_bytes / kernel_ms / 1e6,
            "algorithmic_bytes_per_iteration_per_gpu": total_alg_bytes / args.steps,
            "kernel_time_fraction_of_wall": kernel_ms / 1e3 / dt,
            "kernels": {k: dict(launches=v["launches"], avg_us=v["total_ms"] / v["launches"] * 1e3,
                                gbps=v["bytes_per_launch"] / (v["total_ms"] / v["launches"]) / 1e6,
                                bytes_per_launch=v["bytes_per_launch"]) for k, v in prof.items()},
            "kernel_family": s.kernel_family(),
            "roofline": {"bound": "hbm", "kernel": s.kernel_family().split(" ")[0] + "<" + kname + ">", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "avg_launch_us": avg_ms * 1e3,
                         "algorithmic_bytes_per_launch": kv["bytes_per_launch"]},
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "c5" and n == 10**8:
            out["cpu_baseline"] = cpu_baseline_child(3 * 10**7, n, c1, c2, False)          # ≈ 10–15 s of CPU work
            out["cpu_baseline_all_cores"] = cpu_baseline_child(3 * 10**7, n, c1, c2, True)
        print(json.dumps(out))
    s.close()
    obj.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
