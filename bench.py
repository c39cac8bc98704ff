#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          # N > 1: spawns the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE outer iteration of minimizeobjective (reference src/engine/optim.jl:50-160:
line search + getβ + iterate/direction update) on BASELINE config 5: separable quadratic
f = ½ Σ D_i x_i², D_i = 1 + 999·U_i (counter RNG, seed 24), n = 1e8, x0 = 1, Polak–Ribière β,
StrongWolfeBisection(c1 = 1e-5, c2 = 0.1).  With N > 1 the SAME n = 1e8 state vector is
sharded contiguously over the N GPUs (strong scaling); every fused launch ends in one exchange of
its scalar block (≤ 56 doubles per rank).  For N > 1 the workload is timed on EVERY transport that
passes a sharded self-test — the host shared-memory mailbox the finalize kernels publish into, then the
library's RCCL all-gather over xGMI (the one BASELINE.json's north_star names; run under a watchdog so
that a communicator that never comes up cannot take the measured result down with it) — and both results
are printed (`transports`, with the ranks each transport itself reports and the per-rank exchange wait);
`value` is the faster one and `config.comm` says which.  Inputs are generated on the device and are
resident in HBM before the timed region starts.

Timing: W warm-up steps, then R windows of EXACTLY K steps each, every window bracketed by a barrier
+ torch.cuda.synchronize() on both sides, MAX over ranks.  `value` / `ms_per_step` are the FIRST
window (iterations W+1 … W+K, the contract); `value_median/min/max` summarise all R windows.

Prints ONE JSON line on rank 0 (contract + `roofline` + `cpu_baseline`).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
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


# what the oracle runs for each workload: (objective, β, line search, x0, (warm-up, timed) iterations, sample size)
def _oracle_workload(O, np, workload: str, n: int):
    if workload in ("c5", "c2"):
        return (O.objective("quad_diag", D=O.fill_uniform(n, 24, 1.0, 1000.0)), O.beta_config("PolakRibiere"),
                O.strong_wolfe(1e-5, 0.1), np.ones(n))
    if workload in ("c1", "c1c"):
        return (O.objective("rosenbrock_paired" if workload == "c1" else "rosenbrock_chained"), O.beta_config("PolakRibiere"),
                O.strong_wolfe(1e-5, 0.1), np.tile([-1.2, 1.0], n // 2))
    if workload == "c3":
        return (O.objective("rosenbrock_paired"), O.beta_config("HagerZhang"),
                O.wolfe_bisection("Wolfe", 1e-3, 0.9, 0.0, 100, 1e12, 50), np.tile([-1.2, 1.0], n // 2))
    if workload == "c4":
        return (O.objective("lse", lam=1e-2 / n), O.beta_config("LBFGS", m=10), O.strong_wolfe(1e-5, 0.9),
                O.fill_uniform(n, 24, -5.0, 5.0))
    raise KeyError(workload)


CPU_SAMPLE = {  # workload → (warm-up iterations, timed iterations, repeats of the whole run): ONE run at the FULL problem size
    "c5": (2, 6, 1), "c2": (4, 60, 1), "c1": (3, 15, 200), "c1c": (3, 15, 200), "c3": (3, 10, 1), "c4": (3, 8, 1),
}
OMP_MIN_N = 100000   # oracle/cgo_oracle.c ORC_OMP_MIN: below it every loop of the OpenMP build runs on the calling thread


def cpu_baseline(workload: str, n: int, all_cores: bool = False):
    """Times the oracle (C restatement of the reference's pass structure; 1 thread, or the -fopenmp build on every host core
    the container may use) on the SAME workload at its FULL size: one run of w + k outer iterations, the oracle stamping
    CLOCK_MONOTONIC at the end of every iteration (orc_results.trace_time) — iterations w+1 … w+k are what is reported.
    No sample of the vector, no scaling by n."""
    import numpy as np
    from oracle import oracle as O
    threaded = all_cores and n >= OMP_MIN_N     # (a problem below the oracle's OpenMP threshold runs on one thread in either build)
    if threaded and "OMP_NUM_THREADS" not in os.environ:
        os.environ["OMP_NUM_THREADS"] = str(usable_cores())   # read by libgomp when the OpenMP build is loaded
    if threaded:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # idle team members must not burn the container's CPU quota
    O.use_openmp(threaded)
    obj, beta, ls, x0 = _oracle_workload(O, np, workload, n)
    w, k, reps = CPU_SAMPLE[workload]
    dts, r = [], None
    if reps > 1:   # tiny problems (a solve takes ≈ 0.1 ms): spin the core up first — measured on the GPU boxes' EPYC 9575F, the same
        t_end = time.perf_counter() + 0.5   # single-thread code ran 31 k it/s in a fresh child process and 134 k once the core had clocked up
        while time.perf_counter() < t_end:
            O.minimizeobjective(obj, x0, O.cg_config(1e-200, beta, w + k, True), ls)
    for _ in range(reps + (1 if reps > 1 else 0)):   # (tiny problems: one untimed run first — page-in, caches)
        r = O.minimizeobjective(obj, x0, O.cg_config(1e-200, beta, w + k, True), ls)
        if r.iters_ran < w + k:
            raise RuntimeError(f"oracle stopped after {r.iters_ran} iterations ({r.status})")
        dts.append(float(r.trace_time[w + k - 1] - r.trace_time[w - 1]))
    if reps > 1:
        dts = dts[1:]
    dt = sum(dts) / len(dts)
    its = k / dt
    evals = float(r.trace_objective_evals[w:].mean())
    O.use_openmp(False)
    cores = (int(os.environ.get("OMP_NUM_THREADS", 0)) or usable_cores()) if threaded else 1
    # the host beside the number: CPU model and a one-thread STREAM triad (numpy, 3 × 128 MB, best of 3)
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    m = 16 * 1024 * 1024
    ta, tb, tc = np.zeros(m), np.ones(m), np.full(m, 2.0)
    best = float("inf")
    for _ in range(3):
        t = time.perf_counter()
        np.multiply(tc, 3.0, out=ta)
        np.add(ta, tb, out=ta)
        best = min(best, time.perf_counter() - t)
    triad = 5 * 8 * m / best / 1e9    # (numpy runs the triad as two passes: 2 reads + 1 write, then 2 reads + 1 write in place → 5 streams counted)
    return dict(value=its, unit="iterations/s", cores=cores, kind="port",
                sample=(f"oracle/cgo_oracle.c (faithful pass structure, {cores} thread{'s' if cores > 1 else ''}"
                        f"{'; the problem is below the OpenMP threshold of the all-cores build' if all_cores and not threaded else ''}) on the same workload "
                        f"({workload}) at its full size n={n:.0e}, outer iterations {w + 1}..{w + k} of one run, timed inside the oracle"
                        f"{' (mean of %d runs)' % reps if reps > 1 else ''} ({evals:.2f} trials/iter, {dt:.2f} s)"),
                host_cores_available=usable_cores(), host_cores_machine=os.cpu_count(), host_cpu_model=model,
                host_stream_triad_gbps_one_thread=round(triad, 1))


def cpu_baseline_child(workload: str, n: int, all_cores: bool):
    """Runs cpu_baseline in a fresh child process: libgomp reads OMP_NUM_THREADS once, when it is first
    loaded — in this process torch has loaded it long before, and its default (every core of the machine,
    256 on the GPU boxes) oversubscribes the container's CPU quota 16-fold (measured: 0.6–0.9 it/s against
    14.9 with 16 threads).  The child never touches the GPU."""
    env = dict(os.environ)
    if all_cores:
        env["OMP_NUM_THREADS"] = str(usable_cores())
    code = ("import json, sys; sys.path.insert(0, %r); import bench; "
            "print(json.dumps(bench.cpu_baseline(%r, %d, all_cores=%r)))" % (ROOT, workload, n, all_cores))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError("cpu_baseline child failed: " + r.stderr[-400:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as fresh child
    processes (python -m torch.distributed.run, one rank per GPU) BEFORE anything in this process has touched
    the GPU, relay their output, return their exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes needs it on this pool)
    env.setdefault("OMP_NUM_THREADS", "1")
    print(f"[bench] spawning {args.gpus} ranks: {' '.join(cmd[1:8])} …", file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    line = None
    for ln in p.stdout:
        t = ln.strip()
        if t.startswith("{"):
            line = t
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("[bench] ranks exited cleanly but printed no result line", file=sys.stderr)
        rc = 1
    return rc


WORKLOADS = {  # workload → (default n, default β, description)
    "c1": (1e3, "PolakRibiere", "extended (paired) Rosenbrock n=1000, x0=(-1.2,1,...), StrongWolfeBisection(c1=1e-5,c2=0.1,growth=2) "
                                "[BASELINE config 1 on the GPU engine; latency-bound]"),
    "c1c": (1e3, "PolakRibiere", "chained Rosenbrock n=1000 (stencil objective), x0=(-1.2,1,...), StrongWolfeBisection(c1=1e-5,c2=0.1) "
                                 "[BASELINE config 1, chained form]"),
    "c5": (1e8, "PolakRibiere", "separable quadratic f=1/2 sum D_i x_i^2, D_i=1+999*U_i (splitmix64 counter RNG, seed 24), x0=1, "
                                "StrongWolfeBisection(c1=1e-5,c2=0.1,growth=2) [BASELINE config 5]"),
    "c2": (1e6, "PolakRibiere", "separable quadratic f=1/2 sum D_i x_i^2, D_i=1+999*U_i (seed 24), x0=1, "
                                "StrongWolfeBisection(c1=1e-5,c2=0.1,growth=2) [BASELINE config 2]"),
    "c3": (1e7, "HagerZhang", "extended (paired) Rosenbrock, x0=(-1.2,1,...), WolfeBisection(Wolfe(1e-3,0.9),100,1e12,50) "
                              "[BASELINE config 3]"),
    "c4": (1e7, "LBFGS", "log-sum-exp f=log sum exp(x_i)+lambda/2|x|^2, lambda=1e-2/n, x0_i=5(2U_i-1) (seed 24), L-BFGS m=10, "
                         "StrongWolfeBisection(c1=1e-5,c2=0.9) [BASELINE config 4]"),
}


def git_head() -> str:
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip()
    except Exception:
        return ""


def pmc_traffic(build_id: str, symbol: str, n_local: int):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary — only if that summary was
    collected on THIS build of the library (cgo_build_id), on THIS kernel instantiation and at THIS problem size.
    → (bytes | None, note)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, "no PMC summary under profiles/"
    try:
        d = json.load(open(files[-1]))
    except Exception as e:
        return None, f"unreadable {os.path.basename(files[-1])}: {e}"
    meta = d.get("_meta", {})
    if meta.get("library_build_id") != build_id:
        return None, (f"{os.path.basename(files[-1])} was collected on library build {meta.get('library_build_id')}, "
                      f"this run is build {build_id}: not carried over")
    for k, v in d.items():
        if k != "_meta" and v.get("kernel_symbol") == symbol and int(v.get("n_local", -1)) == int(n_local):
            return v.get("hbm_bytes_per_launch"), (f"{os.path.basename(files[-1])} entry {k} (same build, {symbol}, n={n_local:.0e}, git {meta.get('git_head')}; "
                                                   f"separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled)")
    return None, f"{os.path.basename(files[-1])} holds no entry for {symbol} at n={n_local}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=5, help="R: timed windows of --steps steps each (value = the first)")
    ap.add_argument("--size", dest="n", type=float, default=None, help="global problem size (default: the workload's)")
    ap.add_argument("--workload", default="c5", choices=sorted(WORKLOADS),
                    help="BASELINE.json config: c5 (default, headline) quadratic n=1e8 PR-CG; c1 / c1c Rosenbrock n=1000 PR-CG (paired / "
                         "chained; plain PR leaves the descent cone after 25 iterations on the paired form: use --steps 15 --windows 1); c2 quadratic n=1e6 PR-CG; "
                         "c3 extended Rosenbrock n=1e7 HZ + WolfeBisection; c4 log-sum-exp n=1e7 L-BFGS m=10")
    ap.add_argument("--beta", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-placement-search", action="store_true",
                    help="leave cgo_solver_policy.placement_search at the library's default (off): the launch runs on the buffers as allocated")
    ap.add_argument("--comm", default="auto", choices=["auto", "shm", "rccl", "torch"],
                    help="N > 1 scalar exchange: auto = time the workload on the library's RCCL communicator AND on the host "
                         "shared-memory mailbox, headline = the faster; shm / rccl / torch = that transport only")
    ap.add_argument("--transport-timeout", type=float, default=240.0,
                    help="N > 1: seconds a further transport may take (set-up + self-test + timed windows) once one transport "
                         "has been measured; past it the measured ones are reported and the run ends (0 = wait forever)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for rendezvous/barriers (gloo: rehearsal with several ranks on one GPU)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))   # nothing has touched the GPU yet: torch is imported below

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch  # device memory plumbing / torch.distributed only; loads the HIP runtime first
    import torch.distributed as dist
    import numpy as np
    import cgo_amd as cgo

    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    on_gpu = args.backend == "nccl"
    tdev = "cuda" if on_gpu else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")

    def log(msg):
        print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    def agree(ok) -> bool:
        """MIN over ranks of a local success flag.  EVERY rank calls this the same number of times, in the same
        order, whatever happened locally (local failures only lower the flag) — so the collectives always match."""
        if world == 1:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    W = WORKLOADS[args.workload]
    n = int(args.n if args.n else W[0])
    bname = args.beta or W[1]
    c1, c2 = 1e-5, 0.1
    R = max(1, args.windows)
    beta = {"PolakRibiere": cgo.PolakRibiere(), "HagerZhang": cgo.HagerZhang(), "DaiYuan": cgo.DaiYuan(),
            "LBFGS": cgo.LBFGS(10)}[bname]
    cfg = cgo.setupCGConfig(1e-200, beta, cgo.EnableTrace(), max_iters=args.warmup + R * args.steps + 8)

    # the one policy this benchmark chooses itself: the buffer placement search of DESIGN.md §2.5 (opt-in since round 4; the line
    # reports what it found and at which level the launch runs).  --no-placement-search: the library's default.
    pol = None if args.no_placement_search else cgo.SolverPolicy(placement_search=True)

    def make_solver(ctx):
        if args.workload in ("c5", "c2"):
            obj = cgo.QuadDiagRandom(n, 24, 1.0, 1000.0, ctx)
            s = cgo.Solver(obj, cfg, cgo.setupStrongWolfeBisection(c1, c2), pol)
            s.set_x0_fill("constant", 1.0)
        elif args.workload in ("c1", "c1c"):
            obj = cgo.RosenbrockPaired(n, ctx) if args.workload == "c1" else cgo.RosenbrockChained(n, ctx)
            s = cgo.Solver(obj, cfg, cgo.setupStrongWolfeBisection(c1, c2), pol)
            s.set_x0_fill("alternate", -1.2, 1.0)
        elif args.workload == "c3":
            obj = cgo.RosenbrockPaired(n, ctx)
            s = cgo.Solver(obj, cfg, cgo.WolfeBisection(cgo.Wolfe(1e-3, 0.9), 100, 1e12, 50), pol)
            s.set_x0_fill("alternate", -1.2, 1.0)
        else:
            obj = cgo.LogSumExp(n, 1e-2 / n, ctx)
            s = cgo.Solver(obj, cfg, cgo.setupStrongWolfeBisection(1e-5, 0.9), pol)
            s.set_x0_fill("uniform", -5.0, 5.0, seed=24)
        return obj, s

    # ------------------------------------------------------------------ transports (N > 1)
    devmail = {}

    def torch_allgather(send):
        t = torch.from_numpy(send.copy())
        if on_gpu:
            t = t.cuda()
        out = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy()

    def setup_transport(kind, ctx) -> bool:
        """Collective calls are unconditional; local failures only lower `ok`."""
        ok = True
        if kind == "shm":
            box = [f"/cgo_bench_{os.getpid()}_{int(time.time() * 1e6) & 0xFFFFFF}" if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            if rank == 0:
                try:
                    ctx.set_comm_shm(rank, world, box[0], True)
                except Exception as e:
                    log(f"shared-memory mailbox: create failed ({e})"); ok = False
            dist.barrier()
            if rank != 0:
                try:
                    ctx.set_comm_shm(rank, world, box[0], False)
                except Exception as e:
                    log(f"shared-memory mailbox: open failed ({e})"); ok = False
            dist.barrier()
            if rank == 0:
                try:
                    cgo.shm_unlink(box[0])
                except Exception:
                    pass
            if agree(ok):     # collective: every rank opens its peers' device mailboxes (hipIpc over xGMI); False = host mailbox only
                try:
                    devmail["shm"] = ctx.connect_devices()
                except Exception as e:
                    log(f"device mailboxes: {e}"); devmail["shm"] = False
                return agree(True)
            return False
        if kind == "rccl":
            force = os.environ.get("CGO_BENCH_TRY_RCCL") == "1"
            if not agree((on_gpu or force) and cgo.rccl_available()):   # ncclCommInitRank is collective: all or nobody
                return False
            uid = None
            if rank == 0:
                try:
                    uid = cgo.comm_unique_id()
                except Exception as e:
                    log(f"ncclGetUniqueId failed ({e})")
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            if not agree(box[0] is not None):
                return False
            try:
                ctx.set_comm_rccl(rank, world, box[0])
            except Exception as e:
                log(f"RCCL communicator failed ({e})"); ok = False
            return agree(ok)
        ctx.set_comm_callback(rank, world, torch_allgather)
        return agree(True)

    def selftest(ctx) -> bool:
        """A tiny sharded solve: every rank must finish it and hold the same objective, bit for bit."""
        f, ok = float("nan"), True
        try:
            o = cgo.QuadDiagRandom(8192, 24, 1.0, 1000.0, ctx)
            s0 = cgo.Solver(o, cgo.setupCGConfig(1e-200, cgo.DaiYuan(), cgo.EnableTrace(), max_iters=4),
                            cgo.setupStrongWolfeBisection(1e-5, 0.1))
            s0.set_x0_fill("constant", 1.0); s0.start(); s0.iterate(4)
            f = s0.results(vectors=False).objective
            s0.close(); o.close()
        except Exception as e:
            log(f"exchange self-test failed: {e}"); ok = False
        if world > 1:   # outside the try block: every rank always joins this reduction
            t = torch.tensor([f, -f] if ok and f == f else [float("inf"), float("inf")], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ok = ok and float(t[0].item()) == f and float(t[1].item()) == -f
        return agree(ok)

    # ------------------------------------------------------------------ one timed run on one context
    def timed_run(ctx, label):
        """→ dict of this transport's results (rank 0 fills the kernel table), or None if any rank failed."""
        res, ok = None, True
        obj = s = None
        try:
            obj, s = make_solver(ctx)
            s.start()
            if args.warmup > 0:
                s.iterate(args.warmup)
            no_prof = os.environ.get("CGO_BENCH_NO_PROFILE") == "1"   # experiments only: timing without the HIP-event ring
            s.profile(not no_prof)
            s.profile_reset()
            ctx.exchange_stats(reset=True)
            ctl0 = s.controller_launches()
        except Exception as e:
            log(f"{label}: setup failed: {e}"); ok = False
        if not agree(ok):
            return None
        windows, finished = [], False
        for _ in range(R):
            barrier()
            t0 = time.perf_counter()
            try:
                finished = s.iterate(args.steps)
            except Exception as e:
                log(f"{label}: iterate failed: {e}"); ok = False
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=tdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            if not agree(ok):
                return None
            if finished:      # the solve reached a terminal status inside this window (replicated control flow: every rank
                break         # sees the same): the window is not K full steps — drop it, keep the complete ones
            windows.append(dt)
        xw = [0.0, 0.0]
        try:
            prof = s.profile_get()
            r = s.results(vectors=False)
            nw = len(windows)
            if nw == 0:
                raise RuntimeError(f"solver stopped inside the first timed window: status={r.status} after {r.iters_ran} iterations")
            xn, xpw, xdev = ctx.exchange_stats()
            kind, seen = ctx.comm_info()
            vals = [args.steps / w for w in windows]
            dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])[0] if prof else None
            res = dict(value=vals[0], ms_per_step=windows[0] / args.steps * 1e3,
                       value_median=statistics.median(vals), value_min=min(vals), value_max=max(vals), windows=nw,
                       wall_s=sum(windows), comm=kind, n_ranks_seen=seen,
                       trials_per_iteration=float(r.trace.objective_evals[args.warmup:args.warmup + args.steps].mean()),
                       trials_per_iteration_all_windows=float(r.trace.objective_evals[args.warmup:].mean()),
                       launches_per_iteration=sum(v["launches"] for v in prof.values()) / max(r.iters_ran - args.warmup, 1) if prof else None,
                       controller_armed_launches_per_iteration=(s.controller_launches() - ctl0) / max(r.iters_ran - args.warmup, 1),
                       iters_timed=r.iters_ran - args.warmup, stopped_early=(r.status if finished else None),
                       exchanges=xn, prof=prof, n_per_gpu=obj.n_local, kernel_family=s.kernel_family(),
                       profile="off" if no_prof else "on", dominant=dom, device_mailboxes=devmail.get(label),
                       dominant_symbol=s.kernel_symbol(dom) if dom else "")
            xw = [xpw / max(xn, 1), xdev]
            pf, pb, pc = s.placement_info()
            if pc:      # the placement search of DESIGN.md §2.5 ran: what it found, and at which of the measured LEVELS the launch now runs
                tb = 40.0 * obj.n_local / pb / 1e6 if pb > 0 else 0.0   # TB/s of the bare 5-stream mix on the buffers in use
                res["placement"] = dict(candidates=pc, mix_as_allocated_us=pf, mix_chosen_us=pb, mix_chosen_tbps=tb,
                                        level=("fast" if tb >= 6.1 else ("middle" if tb >= 5.7 else "slow")),
                                        note=(None if tb >= 6.1 else f"no fast (x, u, D) buffer triple among the {pc} candidates of this process: "
                                              "the launch runs at what THESE buffers give the bare stream mix"))
            else:
                res["placement"] = None
            if rank == 0 and world == 1 and dom == "accept_dir_trial" and args.workload in ("c5", "c2") and obj.n_local >= 3 * 10**7:
                if pc:     # the bare mix on the very buffers the solver runs on (timed during its placement search)
                    res["mix_ceiling_us"] = pb
                else:
                    s.close(); obj.close(); s = None      # free the solver's vectors first: the harness allocates three of its own
                    res["mix_ceiling_us"] = cgo.bench_stream_mix(res["n_per_gpu"], 9, ctx)[0]
        except Exception as e:
            log(f"{label}: collecting results failed: {e}"); ok = False
        if world > 1:   # per-rank exchange cost → every rank (always joined, whatever happened above)
            t = torch.tensor(xw, dtype=torch.float64, device=tdev)
            allv = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allv, t)
            if res is not None:
                res["exchange_wait_us_per_launch"] = [round(float(v[0].item()), 2) for v in allv]
                res["exchange_device_us_per_launch"] = [round(float(v[1].item()), 2) for v in allv]
        if s is not None:
            try:
                s.close(); obj.close()
            except Exception:
                pass
        return res if agree(ok) else None

    def emit(results, note=None):
        """rank 0 prints THE result line from whatever transports have completed; returns False if none has."""
        good = {k: v for k, v in results.items() if v}
        if not good:
            return False

        if rank == 0:
            best = max(good, key=lambda k: good[k]["value_median"])
            b = good[best]
            prof = b["prof"]
            names = {"none": "none", "rccl": "rccl all-gather over xGMI (library communicator)",
                     "shm": "host shared-memory mailbox (each rank's launch publishes its block into a POSIX shm segment)",
                     "torch": "torch.distributed callback"}
            if b["profile"] == "off":
                print(json.dumps({k: b[k] for k in ("value", "ms_per_step", "value_median", "value_min", "value_max", "profile",
                                                     "controller_armed_launches_per_iteration", "trials_per_iteration")}))
            else:
                kname = b["dominant"]
                kv = prof[kname]
                avg_ms = kv["total_ms"] / kv["launches"]
                achieved = kv["bytes_per_launch"] / avg_ms / 1e6  # GB/s
                hbm = {k: v for k, v in prof.items() if v["bytes_per_launch"] > 0}
                total_alg_bytes = sum(v["bytes_per_launch"] * v["launches"] for v in hbm.values())
                kernel_ms = sum(v["total_ms"] for v in hbm.values())
                wall_s = b["wall_s"]
                build_id = cgo.build_id()
                traffic, traffic_note = (None, "single-GPU runs only")
                if world == 1:
                    traffic, traffic_note = pmc_traffic(build_id, b["dominant_symbol"], b["n_per_gpu"])
                out = {
                    "metric": ("CG iterations/sec at n=1e8 (outer iterations of minimizeobjective, PR-CG)" if args.workload == "c5" and n == 10**8
                               else f"outer iterations/sec of minimizeobjective, workload {args.workload}, n={n:.0e}, {bname}"),
                    "value": b["value"],
                    "unit": "iterations/s",
                    "n_gpus": world,
                    "steps": args.steps,
                    "warmup": args.warmup,
                    "ms_per_step": b["ms_per_step"],
                    "higher_is_better": True,
                    "scaling": "strong",
                    "vs_baseline": None,
                    "dtype": "f64",
                    "data": "synthetic",
                    "config": {
                        "workload": f"n={n:.0e}, {bname}: " + W[2],
                        "n": n,
                        "n_per_gpu": b["n_per_gpu"],
                        "sharding": "contiguous n/N per GPU; one exchange of 10–56 doubles per fused launch" if world > 1 else "single GPU",
                        "comm": names[best],
                        "n_ranks_seen": b["n_ranks_seen"],
                        "trials_per_iteration": b["trials_per_iteration"],
                        "launches_per_iteration": b["launches_per_iteration"],
                        "controller_armed_launches_per_iteration": b["controller_armed_launches_per_iteration"],
                    },
                    "value_median": b["value_median"], "value_min": b["value_min"], "value_max": b["value_max"],
                    "windows": b["windows"], "stopped_early": b["stopped_early"],
                    "achieved_hbm_gbps_per_gpu_all_kernels": total_alg_bytes / kernel_ms / 1e6,
                    "algorithmic_bytes_per_iteration_per_gpu": total_alg_bytes / max(b["iters_timed"], 1),
                    "kernel_time_fraction_of_wall": kernel_ms / 1e3 / wall_s,
                    "kernels": {k: dict(launches=v["launches"], avg_us=v["total_ms"] / v["launches"] * 1e3,
                                        gbps=v["bytes_per_launch"] / (v["total_ms"] / v["launches"]) / 1e6 if v["total_ms"] > 0 else None,
                                        bytes_per_launch=v["bytes_per_launch"]) for k, v in prof.items()},
                    "kernel_family": b["kernel_family"],
                    "library_build_id": build_id, "git_head": git_head(), "host": socket.gethostname(),
                    "roofline": {"bound": "hbm", "kernel": b["dominant_symbol"] or kname, "kernel_kind": kname, "achieved": achieved,
                                 "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                                 "traffic": traffic, "traffic_source": traffic_note, "avg_launch_us": avg_ms * 1e3,
                                 "algorithmic_bytes_per_launch": kv["bytes_per_launch"]},
                }
                if b.get("mix_ceiling_us"):
                    mix_gbps = 40.0 * b["n_per_gpu"] / b["mix_ceiling_us"] / 1e3
                    out["roofline"]["measured_mix_ceiling_gbps"] = mix_gbps
                    out["roofline"]["frac_of_measured_mix"] = achieved / mix_gbps
                    out["roofline"]["mix_ceiling_source"] = (
                        "k_stream_mix (R x,u,D / W x,u in place without arithmetic, same streaming policy) on the solver's OWN buffers, timed during its placement search"
                        if b.get("placement") else
                        "cgo_bench_stream_mix on this box, right after the timed region: R x,u,D / W x,u in place without arithmetic, same streaming policy, "
                        "median of 9 launches on freshly allocated buffers")
                out["placement"] = b.get("placement")
                out["roofline"]["placement_level"] = (b.get("placement") or {}).get("level")
                if world > 1:
                    out["transports"] = {k: {kk: v[kk] for kk in ("value", "ms_per_step", "value_median", "value_min", "value_max", "comm",
                                                                  "n_ranks_seen", "device_mailboxes", "trials_per_iteration", "launches_per_iteration",
                                                                  "exchanges", "exchange_wait_us_per_launch", "exchange_device_us_per_launch")}
                                         for k, v in good.items()}
                    out["transports_failed"] = [k for k, v in results.items() if not v]
                    # the curve per transport at a glance (north_star names RCCL; `value` is the faster of the two)
                    out["value_rccl"] = good["rccl"]["value"] if "rccl" in good else None
                    out["value_shm"] = good["shm"]["value"] if "shm" in good else None
                if world == 1 and not args.no_cpu_baseline:      # the oracle on the SAME workload, beside every line (≈ 5–30 s of CPU work per leg)
                    for key, allc in (("cpu_baseline", False), ("cpu_baseline_all_cores", True)):
                        try:
                            out[key] = cpu_baseline_child(args.workload, n, allc)
                        except Exception as e:
                            out[key] = {"value": None, "unit": "iterations/s", "cores": None, "kind": "port", "sample": f"failed: {e}"}
                    one, allc = out.get("cpu_baseline", {}), out.get("cpu_baseline_all_cores", {})
                    if one.get("value") and allc.get("value") and allc["value"] < one["value"]:
                        # a competent CPU user takes the faster build: where the threads lose at this size on this host, the all-cores
                        # figure IS the one-thread figure — and the line says so
                        out["cpu_baseline_all_cores"] = dict(one, sample=one["sample"] + f" [the OpenMP build on {allc['cores']} threads was slower "
                                                             f"at this size on this host: {allc['value']:.4g} it/s]")
                if note:
                    out["note"] = note
                print(json.dumps(out), flush=True)
        return True

    results, hung = {}, False
    if world == 1:
        ctx = cgo.Context(dev_index)
        results["none"] = timed_run(ctx, "single GPU")
        ctx.close()
    else:
        # The host mailbox first (no collective library involved: it cannot hang in one), then RCCL — the transport
        # BASELINE.json's north_star names — under a watchdog: should its communicator or its first collective never
        # return on this node, every rank still ends with the result line of what HAS been measured instead of
        # taking the whole scaling run down with it.
        import threading
        order = {"auto": (["shm", "rccl"] if on_gpu else ["shm"]), "shm": ["shm"], "rccl": ["rccl"], "torch": ["torch"]}[args.comm]
        if os.environ.get("CGO_BENCH_TRY_RCCL") == "1" and "rccl" not in order:
            order.append("rccl")   # rehearsal of the failure paths: several ranks on ONE GPU, where RCCL cannot come up

        def give_up(kind):
            # EXIT CODE 5 = "a transport that was attempted hung": the result line of the transports that DID complete is still
            # printed first (with the hung one listed under transports_failed), but the run does not pass for a clean one.
            log(f"transport {kind}: no progress within {args.transport_timeout:.0f} s — reporting the transports measured so far, exit code 5")
            results[kind] = None
            if rank == 0:
                emit(results, note=f"transport '{kind}' timed out after {args.transport_timeout:.0f} s: listed in transports_failed, exit code 5")
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(5)

        for kind in order:
            dog = None
            if any(results.values()) and args.transport_timeout > 0:
                dog = threading.Timer(args.transport_timeout, give_up, args=(kind,))
                dog.daemon = True
                dog.start()
            ctx = cgo.Context(dev_index)
            good = False
            try:
                good = setup_transport(kind, ctx) and selftest(ctx)
            except Exception as e:   # a failed collective of torch.distributed itself: nothing left to agree on
                log(f"{kind}: negotiation failed: {e}")
                if dog is not None:
                    dog.cancel()
                    give_up(kind)
                raise SystemExit(3)
            if good:
                results[kind] = timed_run(ctx, kind)
                if results[kind] is None:
                    hung = True       # a rank failed mid-run: its stream may be stuck in a collective — do not destroy
            else:
                results[kind] = None
                if rank == 0:
                    log(f"transport {kind}: unavailable or failed its sharded self-test — skipped")
            if dog is not None:
                dog.cancel()
            if not hung:
                ctx.close()
            else:
                break
        if not any(results.values()) and args.comm == "auto" and not hung:   # last resort: exchange through torch.distributed itself
            ctx = cgo.Context(dev_index)
            if setup_transport("torch", ctx) and selftest(ctx):
                results["torch"] = timed_run(ctx, "torch")
            ctx.close()
    if rank == 0:
        if not emit(results):
            log("no transport completed the workload")
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(4)
    elif not any(results.values()):
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(4)
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass
    # The transport north_star names must not fail silently: RCCL attempted on real GPUs (one rank per device) and not
    # finished → exit code 5 after the line (which lists it under transports_failed).  A rehearsal with several ranks on
    # one device (--backend gloo), where RCCL cannot come up by construction, is exempt.
    rccl_failed = world > 1 and on_gpu and "rccl" in results and not results["rccl"]
    if hung:   # a context was left alive on purpose (its stream may never drain): skip destructors
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(5 if rccl_failed else 0)
    if rccl_failed:
        sys.stdout.flush(); sys.stderr.flush()
        raise SystemExit(5)


if __name__ == "__main__":
    main()
