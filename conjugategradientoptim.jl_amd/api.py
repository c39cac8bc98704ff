"""Host-side mirror of ConjugateGradientOptim.jl's interface for the hot path.

Same names, argument meaning and error behaviour as the reference module
(src/ConjugateGradientOptim.jl:23-29 exports + the qualified names used by
examples/min.jl and examples/constrained.jl), so that parity tests read like
the reference's own usage:

    config = setupCGConfig(1e-5, HagerZhang(), EnableTrace(); max_iters=1000)
    ls     = setupStrongWolfeBisection(1e-5, 0.8; a_max_growth_factor=2.0, ...)
    ret    = minimizeobjective(fdf!, x0, config, ls)
    ret.minimizer, ret.objective, ret.status, ret.trace.objective_evals ...

The one deliberate difference: `fdf!` is a *device objective descriptor*
(QuadDiag, RosenbrockPaired, Booth, ...) instead of a host closure, because the
element-wise f/∇f runs inside the fused HIP kernels.  Passing a Python callable
raises TypeError — there is no CPU path in this package.

All numerical work happens in lib/libcgo_hip.so behind include/cgo.h; this file
only marshals arguments.  The Julia binding (julia/ConjugateGradientOptimAMD.jl)
is the same thin layer over the same symbols.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import BetaConfig, CGConfigC, LSConfigC, LSSConfigC, ResultsC, check, dp, i64p

# ---------------------------------------------------------------------------
# trace traits (src/types.jl:9-11)
# ---------------------------------------------------------------------------


class TraceTrait:
    pass


class EnableTrace(TraceTrait):
    def __repr__(self):
        return "EnableTrace()"


class DisableTrace(TraceTrait):
    def __repr__(self):
        return "DisableTrace()"


# ---------------------------------------------------------------------------
# βConfig subtypes (src/types.jl:5-7, src/cg_flavours.jl, src/qn_flavours.jl)
# ---------------------------------------------------------------------------


class βConfig:
    _kind = -1

    def _c(self) -> BetaConfig:
        return BetaConfig(self._kind, 0, 0.0)

    def __repr__(self):
        return type(self).__name__ + "()"


class CGβConfig(βConfig):
    pass


class QNβConfig(βConfig):
    pass


class HagerZhang(CGβConfig):  # cg_flavours.jl:83
    _kind = 0


class YuanWangSheng(CGβConfig):  # cg_flavours.jl:46-48
    _kind = 1

    def __init__(self, μ: float):
        self.μ = float(μ)

    def _c(self):
        return BetaConfig(self._kind, 0, self.μ)

    def __repr__(self):
        return f"YuanWangSheng({self.μ})"


class SallehAlhawarat(CGβConfig):  # cg_flavours.jl:130
    _kind = 2


class LiuStorrey(CGβConfig):  # cg_flavours.jl:154
    _kind = 3


class PolakRibiere(CGβConfig):  # new (stub at cg_flavours.jl:173-174)
    _kind = 4


class HestenesStiefel(CGβConfig):  # new (commented at cg_flavours.jl:110-127)
    _kind = 5


class DaiYuan(CGβConfig):  # new
    _kind = 6


class LBFGS(QNβConfig):  # new QNβConfig behind the contract of qn_flavours.jl:5-48
    _kind = 7

    def __init__(self, m: int = 10):
        self.m = int(m)

    def _c(self):
        return BetaConfig(self._kind, self.m, 0.0)

    def __repr__(self):
        return f"LBFGS({self.m})"


class BroydenFamily(QNβConfig):
    """BroydenFamily{T}(θ, B)  (src/qn_flavours.jl:53-66).  The reference's update computes s = B\\y first
    (:81), so Bs = y, sBs = s·y, v = 0 and B_new = B up to rounding (:83-87): B stays the identity it is
    initialised to and the direction B\\(−g) is steepest descent for every θ.  The engine therefore runs
    u = −g exactly (no n×n matrix, no O(n³) solves); the oracle carries the dense algebra."""
    _kind = 8

    def __init__(self, θ: float):
        self.θ = float(θ)

    def _c(self):
        return BetaConfig(self._kind, 0, self.θ)

    def __repr__(self):
        return f"BroydenFamily({self.θ})"


def setupBroydenFamily(θ: float, N: int) -> BroydenFamily:
    """setupBroydenFamily(θ, N)  (qn_flavours.jl:55-66): `@assert zero(T) <= θ`; the N×N matrix is never built."""
    if not (0.0 <= θ):
        raise AssertionError("AssertionError: zero(T) <= θ  (qn_flavours.jl:57)")
    return BroydenFamily(θ)


# ---------------------------------------------------------------------------
# CGConfig (src/types.jl:156-203)
# ---------------------------------------------------------------------------
@dataclass
class CGConfig:
    ϵ: float
    β_config: βConfig
    max_iters: int
    verbose: bool
    trace_status: TraceTrait

    def _c(self) -> CGConfigC:
        return CGConfigC(self.ϵ, self.β_config._c(), self.max_iters, int(self.verbose),
                         1 if isinstance(self.trace_status, EnableTrace) else 0)


def setupCGConfig(ϵ: float, β_config: βConfig, trace_status: TraceTrait, *, max_iters: int = 1000,
                  verbose: bool = False) -> CGConfig:
    """setupCGConfig(ϵ, β_config, trace_status; max_iters=1000, verbose=false)  (types.jl:171-203)."""
    cfg = CGConfig(float(ϵ), β_config, int(max_iters), bool(verbose), trace_status)
    c = cfg._c()
    check(_lib.lib().cgo_check_cg_config(C.byref(c)))  # @assert zero(T) < ϵ < one(T)
    return cfg


# ---------------------------------------------------------------------------
# line-search configs (src/linesearch/nocedal.jl:3-30, wolfe.jl:6-11,213-217,259-262)
# ---------------------------------------------------------------------------
class LineSearchConfig:
    pass


@dataclass
class StrongWolfeBisection(LineSearchConfig):
    c1: float
    c2: float
    a_max_growth_factor: float
    max_iters: int
    zoom_max_iters: int

    def _c(self) -> LSConfigC:
        return LSConfigC(0, 0, self.c1, self.c2, self.a_max_growth_factor, 0.0, 0.0,
                         self.max_iters, self.zoom_max_iters, 0, 0.0)


def setupStrongWolfeBisection(c1: float, c2: float, *, a_max_growth_factor: float = 2.0,
                              max_iters: int = 1000, zoom_max_iters: int = 100) -> StrongWolfeBisection:
    """setupStrongWolfeBisection(c1, c2; ...)  (nocedal.jl:14-30) incl. its @asserts."""
    ls = StrongWolfeBisection(float(c1), float(c2), float(a_max_growth_factor), int(max_iters),
                              int(zoom_max_iters))
    c = ls._c()
    check(_lib.lib().cgo_check_ls_config(C.byref(c)))
    return ls


@dataclass
class Wolfe:  # wolfe.jl:259-262
    c1: float
    c2: float


@dataclass
class YuanWeiLuWolfe:  # wolfe.jl:213-217
    c1: float
    c2: float
    δ1: float


@dataclass
class WolfeBisection(LineSearchConfig):
    """WolfeBisection(condition, max_iters, max_step_size, feasibility_max_iters)  (wolfe.jl:6-11).

    Like the reference's raw constructor this does not validate; the condition's
    @assert (wolfe.jl:233,278) fires when the solve starts."""
    condition: object
    max_iters: int
    max_step_size: float
    feasibility_max_iters: int

    def _c(self) -> LSConfigC:
        cond = self.condition
        if isinstance(cond, YuanWeiLuWolfe):
            return LSConfigC(1, 1, cond.c1, cond.c2, 2.0, cond.δ1, self.max_step_size,
                             self.max_iters, 0, self.feasibility_max_iters, 0.0)
        if isinstance(cond, Wolfe):
            return LSConfigC(1, 0, cond.c1, cond.c2, 2.0, 0.0, self.max_step_size, self.max_iters,
                             0, self.feasibility_max_iters, 0.0)
        raise TypeError("WolfeBisection.condition must be Wolfe or YuanWeiLuWolfe")


@dataclass
class Armijo:  # geometric.jl:159-162
    c1: float


@dataclass
class Backtracking(LineSearchConfig):
    """Backtracking(condition, discount_factor, max_iters, feasibility_max_iters)  (geometric.jl:15-20).

    Restated bug for bug (geometric.jl:77-78,141-144): on :success the previous (ϕ, a) is returned while
    the adopted iterate/gradient are those of the last, rejected trial — exactly what the reference does."""
    condition: Armijo
    discount_factor: float
    max_iters: int
    feasibility_max_iters: int

    def _c(self) -> LSConfigC:
        if not isinstance(self.condition, Armijo):
            raise TypeError("Backtracking.condition must be Armijo")
        return LSConfigC(2, 2, self.condition.c1, 0.0, 2.0, 0.0, 0.0, self.max_iters, 0,
                         self.feasibility_max_iters, self.discount_factor)


def evalwolfeconditions(condition, ϕ_a: float, dϕ_a: float, a: float, u, ϕ_0: float, dϕ_0: float):
    """evalwolfeconditions(condition, ϕ_a, dϕ_a, a, u, ϕ_0, dϕ_0) → (valid_large, valid_small)  (wolfe.jl:219-294).
    `u` is the search direction or — since only YuanWeiLuWolfe reads it, and only as dot(u,u) (:240) — that scalar."""
    uu = float(u) if np.isscalar(u) else float(np.dot(np.asarray(u, dtype=np.float64), np.asarray(u, dtype=np.float64)))
    ls = WolfeBisection(condition, 1, 1.0, 1)._c()
    v1, v2 = C.c_int32(0), C.c_int32(0)
    check(_lib.lib().cgo_evalwolfeconditions(C.byref(ls), ϕ_a, dϕ_a, a, uu, ϕ_0, dϕ_0, C.byref(v1), C.byref(v2)))
    return bool(v1.value), bool(v2.value)


def evalbacktrackcondition(condition: Armijo, ϕ_a: float, a: float, ϕ_0: float, dϕ_0: float) -> bool:
    """evalbacktrackcondition(::Armijo, ϕ_a, a, ϕ_0, dϕ_0)  (geometric.jl:164-186)."""
    ls = Backtracking(condition, 0.5, 1, 1)._c()
    v = C.c_int32(0)
    check(_lib.lib().cgo_evalbacktrackcondition(C.byref(ls), ϕ_a, a, ϕ_0, dϕ_0, C.byref(v)))
    return bool(v.value)


@dataclass
class LinesearchSolveSys:
    """LinesearchSolveSys{T}(ρ, σ, s, max_iters)  (solve_system.jl:6-11): the line search of
    solvesystem — eqn 18 of (Yuan 2019): first s·ρ^i with −g(z)ᵀu ≥ σ·a·‖g(z)‖·‖u‖²."""
    ρ: float
    σ: float
    s: float
    max_iters: int

    def _c(self) -> LSSConfigC:
        return LSSConfigC(self.ρ, self.σ, self.s, self.max_iters)


def setupLinesearchSolveSys(s: float, σ: float = 0.5, ρ: float = 0.95, max_iters: Optional[int] = None) -> LinesearchSolveSys:
    """setupLinesearchSolveSys(s; σ = 0.5, ρ = 0.95, max_iters = round(Int, log(ρ, 1e-6)))  (solve_system.jl:13-27)."""
    if max_iters is None:
        max_iters = int(_lib.lib().cgo_lss_default_max_iters(ρ)) if 0.0 < ρ < 1.0 else 0
    cfg = LinesearchSolveSys(ρ, σ, s, int(max_iters))
    c = cfg._c()
    check(_lib.lib().cgo_check_lss_config(C.byref(c)))   # the @assert's of solve_system.jl:21-23
    return cfg


# ---------------------------------------------------------------------------
# Results / TraceContainer (src/types.jl:17-23,107-114)
# ---------------------------------------------------------------------------
@dataclass
class TraceContainer:
    objective: np.ndarray
    grad_norm: np.ndarray
    step_size: np.ndarray
    objective_evals: np.ndarray
    status: TraceTrait


@dataclass
class Results:
    objective: float
    minimizer: np.ndarray
    gradient: np.ndarray
    iters_ran: int
    status: str  # the reference's Symbol, as text
    trace: TraceContainer
    total_fdf_evals: int = 0
    total_launches: int = 0


# ---------------------------------------------------------------------------
# device context
# ---------------------------------------------------------------------------
class Context:
    """One GPU = one shard.  `Context()` is a single-rank context on device 0."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        check(_lib.lib().cgo_ctx_create(device, C.byref(self._h)))
        self.device = device
        self.rank, self.world = 0, 1
        self._keep = []

    def set_default_policy(self, policy: "Optional[SolverPolicy]"):
        """The policy of every solver this context creates from now on — minimizeobjective / minimizeobjectivererun /
        solvesystem included (cgo_ctx_set_default_policy); None resets to the library's."""
        self._defpol_c = policy._c() if policy is not None else None
        check(_lib.lib().cgo_ctx_set_default_policy(self._h, C.byref(self._defpol_c) if self._defpol_c is not None else None))

    def set_comm_rccl(self, rank: int, world: int, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(_lib.lib().cgo_ctx_set_comm_rccl(self._h, rank, world, buf))
        self.rank, self.world = rank, world

    def set_comm_callback(self, rank: int, world: int, allgather):
        """allgather(send: np.ndarray[count]) -> np.ndarray[world*count] (rank-major)."""
        def tramp(_user, send, recv, count):
            try:
                s = np.ctypeslib.as_array(send, shape=(count,)).copy()
                r = np.asarray(allgather(s), dtype=np.float64).reshape(-1)
                np.ctypeslib.as_array(recv, shape=(world * count,))[:] = r
                return 0
            except Exception:  # pragma: no cover - surfaced as CGO_ECOMM
                return 1
        cb = _lib.ALLGATHER_FN(tramp)
        self._keep.append(cb)
        check(_lib.lib().cgo_ctx_set_comm_callback(self._h, rank, world, cb, None))
        self.rank, self.world = rank, world

    def set_comm_shm(self, rank: int, world: int, name: str, create: bool):
        """Host shared-memory mailbox (one node).  Call on rank 0 with create=True, synchronise,
        call on the other ranks with create=False, synchronise, then rank 0 may shm_unlink(name)."""
        check(_lib.lib().cgo_ctx_set_comm_shm(self._h, rank, world, name.encode(), int(create)))
        self.rank, self.world = rank, world

    def connect_devices(self) -> bool:
        """COLLECTIVE (every rank, after all have attached with set_comm_shm): open the peers' device mailboxes over
        hipIpc / xGMI so that controller-armed launches exchange their blocks GPU to GPU.  False: host mailbox only."""
        ok = C.c_int32(0)
        check(_lib.lib().cgo_ctx_comm_connect_devices(self._h, C.byref(ok)))
        return bool(ok.value)

    def comm_info(self):
        """(transport, ranks_seen): 'none' | 'shm' | 'rccl' | 'callback', and how many ranks the transport itself
        reports (ncclCommCount / mailbox slots that have published / world)."""
        k, r, w, seen = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        check(_lib.lib().cgo_ctx_comm_info(self._h, C.byref(k), C.byref(r), C.byref(w), C.byref(seen)))
        return {0: "none", 1: "shm", 2: "rccl", 3: "callback"}[k.value], seen.value

    def exchange_stats(self, reset: bool = False):
        """(exchanges, peer_wait_us_total, device_exchange_us_mean) since the last reset — cgo_ctx_exchange_stats."""
        cnt, pw, dx = C.c_int64(), C.c_double(), C.c_double()
        check(_lib.lib().cgo_ctx_exchange_stats(self._h, C.byref(cnt), C.byref(pw), C.byref(dx), int(reset)))
        return cnt.value, pw.value, dx.value

    def close(self):
        if self._h:
            _lib.lib().cgo_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_id() -> str:
    """Digest of the sources libcgo_hip.so was built from (cgo_build_id)."""
    return _lib.lib().cgo_build_id().decode()


def rccl_available() -> bool:
    return _lib.lib().cgo_rccl_available() == 1


def shm_unlink(name: str) -> None:
    _lib.lib().cgo_shm_unlink(name.encode())


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    check(_lib.lib().cgo_comm_unique_id(buf))
    return buf.raw


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def shard_extent(n_global: int, rank: int, world: int):
    """Contiguous, even-aligned shard [offset, offset+n_local) of rank `rank`."""
    pairs = n_global // 2
    base, rem = divmod(pairs, world)
    p0 = rank * base + min(rank, rem)
    p1 = p0 + base + (1 if rank < rem else 0)
    off, end = 2 * p0, 2 * p1
    if rank == world - 1:
        end = n_global  # odd tail element goes to the last rank
    return off, end - off


# ---------------------------------------------------------------------------
# device objective descriptors — the GPU-side `fdf!`
# ---------------------------------------------------------------------------
OBJ_KINDS = {"quad_diag": 0, "rosenbrock_paired": 1, "booth": 2, "lse": 3, "rosenbrock_chained": 6}
FILL_KINDS = {"constant": 0, "uniform": 1, "alternate": 2}


class DeviceObjective:
    """Descriptor of an element-wise objective evaluated inside the fused kernels."""

    def __init__(self, kind: str, n_global: int, ctx: Optional[Context] = None, source: Optional[str] = None,
                 has_param: bool = False):
        self.ctx = ctx or default_context()
        self.kind = kind
        self.n_global = int(n_global)
        # Context.shard_fn: a host that partitions differently (uneven shards) supplies its own (n_global, rank, world) → extents
        shard = getattr(self.ctx, "shard_fn", None) or shard_extent
        self.offset, self.n_local = shard(self.n_global, self.ctx.rank, self.ctx.world)
        self._h = C.c_void_p()
        if source is not None:
            check(_lib.lib().cgo_objective_create_from_source(
                self.ctx._h, source.encode(), int(has_param), self.n_global, self.offset, self.n_local,
                C.byref(self._h)))
            return
        check(_lib.lib().cgo_objective_create(self.ctx._h, OBJ_KINDS[kind], self.n_global,
                                              self.offset, self.n_local, C.byref(self._h)))

    def local(self, v: np.ndarray) -> np.ndarray:
        """This rank's shard of a global host vector."""
        return np.ascontiguousarray(v[self.offset:self.offset + self.n_local], dtype=np.float64)

    def set_param(self, v_global: np.ndarray, slot: int = 0):
        loc = self.local(np.asarray(v_global, dtype=np.float64))
        check(_lib.lib().cgo_objective_set_param_host(self._h, slot, loc.ctypes.data_as(dp)))

    def fill_param(self, fill: str, seed: int, lo: float, hi: float, slot: int = 0):
        check(_lib.lib().cgo_objective_fill_param(self._h, slot, FILL_KINDS[fill], seed, lo, hi))

    def set_scalar(self, value: float, slot: int = 0):
        check(_lib.lib().cgo_objective_set_scalar(self._h, slot, value))

    def __call__(self, g: np.ndarray, x: np.ndarray) -> float:
        """f = fdf!(g, x) on host vectors (one launch) — the reference's callback contract."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        f = C.c_double()
        check(_lib.lib().cgo_objective_eval_host(self._h, x.ctypes.data_as(dp), g.ctypes.data_as(dp),
                                                 C.byref(f)))
        return f.value

    def close(self):
        if self._h:
            _lib.lib().cgo_objective_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostObjective(DeviceObjective):
    """The reference's objective contract as it is: a host closure `f = fdf!(g, x)` that writes the gradient into `g`
    in place and returns the value (src/engine/optim.jl:25, src/cg_utils.jl:19; e.g. `boothfdf!`,
    examples/helpers/test_funcs.jl:3-12).  The solve still runs on the GPU engine — state, direction updates, dots,
    norms and the line-search state machine are the device ones; only x + a·u goes out and g comes back around each
    call of the closure (cgo_objective_create_callback).  `minimizeobjective(closure, …)` wraps a plain callable in
    one of these, so the reference's call `minimizeobjective(boothfdf!, x0, config, ls)` is a drop-in."""

    def __init__(self, fdf, n_global: int, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()
        self.kind = "host"
        self.n_global = int(n_global)
        self.offset, self.n_local = shard_extent(self.n_global, self.ctx.rank, self.ctx.world)
        self._h = C.c_void_p()
        self._err = None

        def tramp(_user, gp, xp, n):
            try:
                g = np.ctypeslib.as_array(gp, shape=(n,))
                x = np.ctypeslib.as_array(xp, shape=(n,))
                return float(fdf(g, x))
            except BaseException as e:   # an exception cannot cross the C boundary: NaN stops the solve, re-raised after
                self._err = e
                return float("nan")
        self._cb = _lib.FDF_FN(tramp)
        self._fdf = fdf
        check(_lib.lib().cgo_objective_create_callback(self.ctx._h, self._cb, None, self.n_global, self.offset,
                                                       self.n_local, C.byref(self._h)))

    def reraise(self):
        if self._err is not None:
            e, self._err = self._err, None
            raise e


def QuadDiag(D: np.ndarray, ctx: Optional[Context] = None) -> DeviceObjective:
    """f(x) = ½ Σ D_i x_i²."""
    D = np.asarray(D, dtype=np.float64)
    o = DeviceObjective("quad_diag", D.size, ctx)
    o.set_param(D)
    return o


def QuadDiagRandom(n: int, seed: int = 24, lo: float = 1.0, hi: float = 1000.0,
                   ctx: Optional[Context] = None) -> DeviceObjective:
    """½ Σ D_i x_i² with D_i = lo + (hi−lo)·U(seed ⊕ i) generated on the device."""
    o = DeviceObjective("quad_diag", n, ctx)
    o.fill_param("uniform", seed, lo, hi)
    return o


def RosenbrockPaired(n: int, ctx: Optional[Context] = None) -> DeviceObjective:
    return DeviceObjective("rosenbrock_paired", n, ctx)


def RosenbrockChained(n: int, ctx: Optional[Context] = None) -> DeviceObjective:
    """f(x) = Σ_{i<n−1} (1 − x_i)² + 100 (x_{i+1} − x_i²)² — the chained form of examples/helpers/test_funcs.jl:50-57
    (value there; BASELINE config 1).  A 3-point stencil objective on the device (csrc/cgo_kernels_chain.hip.hpp):
    CG β kinds, every line search; shards exchange a 2-element halo inside the per-launch scalar block."""
    return DeviceObjective("rosenbrock_chained", n, ctx)


def LogSumExp(n: int, λ: float = 0.0, ctx: Optional[Context] = None) -> DeviceObjective:
    """f(x) = log Σ exp(x_i) + ½λ‖x‖² — not element-wise: two-phase kernels (trial statistics, then
    the gradient of the accepted step only)."""
    o = DeviceObjective("lse", n, ctx)
    o.set_scalar(float(λ))
    return o


def ElementwiseObjective(n: int, source: str, param: Optional[np.ndarray] = None,
                         ctx: Optional[Context] = None, cheap: bool = False) -> DeviceObjective:
    """A USER-SUPPLIED element-wise f/∇f — the GPU-side form of passing `minimizeobjective` your own
    `fdf!` closure.  `source` is HIP C++: the statements of an element-wise body setting `fi` and `gi`
    from `x`, `p`, `s0` (e.g. "gi = p*x; fi = 0.5*(gi*x);"), or a full `struct UserObjective {...}`
    functor for pair-coupled objectives.  Compiled at run time (hiprtc) into the fused kernels."""
    o = DeviceObjective("user", n, ctx, source=source, has_param=param is not None)
    if param is not None:
        o.set_param(np.asarray(param, dtype=np.float64))
    if cheap:   # ≲ 10 flops per element for f and ∇f: seven speculative trial steps per launch
        check(_lib.lib().cgo_objective_set_cost_class(o._h, 1))
    return o


def Booth(ctx: Optional[Context] = None) -> DeviceObjective:
    """examples/helpers/test_funcs.jl:3-12."""
    return DeviceObjective("booth", 2, ctx)


def _dev_ptr(t, n: int):
    """torch tensor / raw pointer → void* (None → NULL); checks dtype, contiguity and length of tensors."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    if hasattr(t, "data_ptr"):
        import torch
        if t.dtype != torch.float64 or not t.is_contiguous() or t.numel() != n or not t.is_cuda:
            raise ValueError(f"device vector must be a contiguous float64 GPU tensor of {n} elements")
        return C.c_void_p(t.data_ptr())
    raise TypeError("expected a torch.Tensor on the GPU or a raw device pointer")


# ---------------------------------------------------------------------------
# resumable solver (what minimizeobjective is built from)
# ---------------------------------------------------------------------------
LBFGS_FORMS = {"auto": 0, "one_pass": 1, "one_pass_own_update": 2, "gram": 3, "two_loop": 4}


@dataclass
class SolverPolicy:
    """cgo_solver_policy (include/cgo.h): HOW a solve runs, never WHAT it computes.  `None` = library policy for that
    field.  Precedence per field: Solver(..., policy=) > Context.set_default_policy > CGO_* experiment override > library."""
    points: Optional[int] = None               # trial steps per fused launch: 1 | 3 | 5 | 7
    resident: Optional[bool] = None            # resident solver on / off
    controller_depth: Optional[int] = None     # 0 host-driven, k armed rounds in flight
    controller_graph: Optional[bool] = None
    controller_fused: Optional[bool] = None
    stored_gradient: Optional[bool] = None     # True: the stored-gradient k_fused family
    fused_tail: Optional[bool] = None          # context-wide
    strict_tail: Optional[bool] = None         # context-wide: formally fenced hand-offs
    placement_search: Optional[bool] = None
    placement_stages: Optional[int] = None
    placement_max_bytes: Optional[int] = None  # cap on the search's transient device memory
    lbfgs_form: Optional[str] = None           # "one_pass" | "one_pass_own_update" | "gram" | "two_loop"
    lbfgs_fuse_grad: Optional[bool] = None
    lbfgs_fuse_trial: Optional[bool] = None
    lse_fixed_reference: Optional[bool] = None
    resident_points: Optional[int] = None
    resident_chunk: Optional[int] = None
    hbm_stream_bytes: Optional[float] = None   # launches moving more than this stream pure-HBM style (1: every launch)

    def _c(self) -> "_lib.SolverPolicyC":
        c = _lib.SolverPolicyC()
        _lib.lib().cgo_solver_policy_init(C.byref(c))
        tri = lambda v: -1 if v is None else (1 if v else 0)
        c.points = self.points or 0
        c.resident = tri(self.resident)
        c.controller_depth = -1 if self.controller_depth is None else int(self.controller_depth)
        c.controller_graph, c.controller_fused = tri(self.controller_graph), tri(self.controller_fused)
        c.stored_gradient = 1 if self.stored_gradient else 0
        c.fused_tail, c.strict_tail = tri(self.fused_tail), tri(self.strict_tail)
        c.placement_search = tri(self.placement_search)
        c.placement_stages = self.placement_stages or 0
        c.placement_max_bytes = int(self.placement_max_bytes or 0)
        c.lbfgs_form = LBFGS_FORMS[self.lbfgs_form or "auto"]
        c.lbfgs_fuse_grad, c.lbfgs_fuse_trial = tri(self.lbfgs_fuse_grad), tri(self.lbfgs_fuse_trial)
        c.lse_fixed_reference = tri(self.lse_fixed_reference)
        c.resident_points = self.resident_points or 0
        c.resident_chunk = self.resident_chunk or 0
        c.hbm_stream_bytes = float(self.hbm_stream_bytes or 0.0)
        return c


class Solver:
    def __init__(self, fdf: DeviceObjective, config: CGConfig, linesearch_config: LineSearchConfig,
                 policy: Optional[SolverPolicy] = None):
        if not isinstance(fdf, DeviceObjective):
            raise TypeError(
                "fdf! must be an objective descriptor: a device objective (QuadDiag, RosenbrockPaired, Booth, "
                "ElementwiseObjective, ...) or HostObjective(closure, n) — there is no CPU solver path in this package")
        self.obj, self.config, self.ls = fdf, config, linesearch_config
        self._cfg_c, self._ls_c = config._c(), linesearch_config._c()
        self._h = C.c_void_p()
        # a LinesearchSolveSys config selects solvesystem (solve_system.jl) instead of minimizeobjective
        create = (_lib.lib().cgo_solver_create_sys_ex if isinstance(linesearch_config, LinesearchSolveSys)
                  else _lib.lib().cgo_solver_create_ex)
        self._pol_c = policy._c() if policy is not None else None
        check(create(fdf.ctx._h, fdf._h, C.byref(self._cfg_c), C.byref(self._ls_c),
                     C.byref(self._pol_c) if self._pol_c is not None else None, C.byref(self._h)))

    def policy(self) -> dict:
        """The policy this solver actually runs with (cgo_solver_get_policy), as a dict of the C struct's fields."""
        c = _lib.SolverPolicyC()
        check(_lib.lib().cgo_solver_get_policy(self._h, C.byref(c)))
        return {n: getattr(c, n) for n, _ in c._fields_ if n not in ("size", "reserved")}

    def set_x0(self, x_initial_global: np.ndarray):
        loc = self.obj.local(np.asarray(x_initial_global, dtype=np.float64))
        check(_lib.lib().cgo_solver_set_x0_host(self._h, loc.ctypes.data_as(dp)))

    def set_x0_fill(self, fill: str, lo: float, hi: float = 0.0, seed: int = 0):
        check(_lib.lib().cgo_solver_set_x0_fill(self._h, FILL_KINDS[fill], seed, lo, hi))

    def set_x0_device(self, x0):
        """x_initial from memory that already lives on this GPU: a torch.Tensor (float64, contiguous, this rank's shard)
        or a raw device pointer (int).  Device-to-device copy, no PCIe (cgo_solver_set_x0_device)."""
        check(_lib.lib().cgo_solver_set_x0_device(self._h, _dev_ptr(x0, self.obj.n_local)))

    def results_device(self, minimizer=None, gradient=None):
        """Results.minimizer / Results.gradient into device buffers (torch tensors or raw pointers); scalars and traces
        come from results(vectors=False)."""
        check(_lib.lib().cgo_solver_results_device(self._h, _dev_ptr(minimizer, self.obj.n_local), _dev_ptr(gradient, self.obj.n_local)))

    def start(self):
        check(_lib.lib().cgo_solver_start(self._h))

    def iterate(self, iters: int) -> bool:
        fin = C.c_int32(0)
        check(_lib.lib().cgo_solver_iterate(self._h, iters, C.byref(fin)))
        return bool(fin.value)

    def enable_trial_log(self):
        cnt = C.c_int64()
        check(_lib.lib().cgo_solver_trial_log(self._h, -1, None, None, None, C.byref(cnt)))

    def trial_log(self):
        cnt = C.c_int64()
        check(_lib.lib().cgo_solver_trial_log(self._h, 0, None, None, None, C.byref(cnt)))
        k = cnt.value
        a, p, d = np.zeros(max(k, 1)), np.zeros(max(k, 1)), np.zeros(max(k, 1))
        check(_lib.lib().cgo_solver_trial_log(self._h, k, a.ctypes.data_as(dp), p.ctypes.data_as(dp),
                                              d.ctypes.data_as(dp), C.byref(cnt)))
        return a[:k], p[:k], d[:k]

    def kernel_family(self) -> str:
        return _lib.lib().cgo_solver_kernel_family(self._h).decode()

    def kernel_symbol(self, kind_name: str) -> str:
        """The kernel instantiation launches of kind `kind_name` ('accept_dir_trial', 'trial', …) use under the current
        policy, e.g. 'k_cg<ObjQuadDiag, 7, 7, true>' (cgo_solver_kernel_symbol)."""
        L = _lib.lib()
        buf = C.create_string_buffer(200)
        for k in range(L.cgo_num_kernel_kinds()):
            if L.cgo_kernel_kind_name(k).decode() == kind_name:
                check(L.cgo_solver_kernel_symbol(self._h, k, buf, 200))
                return buf.value.decode()
        return ""

    def placement_info(self):
        """(as_allocated_us, chosen_us, candidates) of the solver's placement search (cgo_solver_placement_info);
        candidates == 0: no search was made."""
        a, b, c = C.c_double(), C.c_double(), C.c_int32()
        check(_lib.lib().cgo_solver_placement_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def controller_launches(self) -> int:
        """Launches armed by the on-device controller instead of the host (csrc/cgo_ctl.hpp)."""
        return int(_lib.lib().cgo_solver_controller_launches(self._h))

    def resident_stats(self):
        """(slices, iterations): launches of the resident solver and the outer iterations completed inside them."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(_lib.lib().cgo_solver_resident_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        self.resident_gave_up = c.value   # slices handed back whole because their workgroups could not all run at once
        return a.value, b.value

    def lbfgs_stats(self):
        """(speculated, fused, plain): how the L-BFGS state updates of this solver were paid for (include/cgo.h)."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(_lib.lib().cgo_solver_lbfgs_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def profile(self, on: bool = True):
        check(_lib.lib().cgo_solver_profile_enable(self._h, int(on)))

    def profile_reset(self):
        check(_lib.lib().cgo_solver_profile_reset(self._h))

    def profile_get(self):
        L = _lib.lib()
        out = {}
        for k in range(L.cgo_num_kernel_kinds()):
            n, ms, b = C.c_int64(), C.c_double(), C.c_double()
            check(L.cgo_solver_profile_get(self._h, k, C.byref(n), C.byref(ms), C.byref(b)))
            if n.value:
                out[L.cgo_kernel_kind_name(k).decode()] = dict(
                    launches=n.value, total_ms=ms.value, bytes_per_launch=b.value)
        return out

    def results(self, vectors: bool = True) -> Results:
        n = self.obj.n_local
        cap = max(self.config.max_iters, 1)
        x = np.empty(n) if vectors else None
        g = np.empty(n) if vectors else None
        to, tg, ts = np.zeros(cap), np.zeros(cap), np.zeros(cap)
        te = np.zeros(cap, dtype=np.int64)
        r = ResultsC()
        r.minimizer = x.ctypes.data_as(dp) if vectors else None
        r.gradient = g.ctypes.data_as(dp) if vectors else None
        r.trace_objective, r.trace_grad_norm = to.ctypes.data_as(dp), tg.ctypes.data_as(dp)
        r.trace_step_size, r.trace_objective_evals = ts.ctypes.data_as(dp), te.ctypes.data_as(i64p)
        check(_lib.lib().cgo_solver_results(self._h, C.byref(r)))
        k = max(int(r.iters_ran), 0)
        if not isinstance(self.config.trace_status, EnableTrace):
            k = 0
        tr = TraceContainer(to[:k].copy(), tg[:k].copy(), ts[:k].copy(), te[:k].copy(),
                            self.config.trace_status)
        return Results(r.objective, x, g, int(r.iters_ran),
                       _lib.lib().cgo_status_name(r.status).decode(), tr,
                       int(r.total_fdf_evals), int(r.total_launches))

    def close(self):
        if self._h:
            _lib.lib().cgo_solver_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
# the reference's two entry points
# ---------------------------------------------------------------------------
def minimizeobjective(fdf, x_initial: Sequence[float], config: CGConfig,
                      linesearch_config: LineSearchConfig, policy: Optional[SolverPolicy] = None) -> Results:
    """minimizeobjective(fdf!, x_initial, config, linesearch_config)  (src/engine/optim.jl:6-171).

    `x_initial` is the GLOBAL initial iterate (copied, never mutated — optim.jl:21);
    the returned minimizer/gradient are this rank's shard (the whole vector on one GPU).
    `fdf` is a device objective descriptor or — the reference's own form — a closure `fdf(g, x) -> f`
    (wrapped in a HostObjective: GPU engine, objective evaluated on the host)."""
    own = None
    if not isinstance(fdf, DeviceObjective) and callable(fdf):
        fdf = own = HostObjective(fdf, len(x_initial))
    s = Solver(fdf, config, linesearch_config, policy)
    try:
        s.set_x0(np.asarray(x_initial, dtype=np.float64))
        s.start()
        while not s.iterate(1 << 40):
            pass
        r = s.results()
        if isinstance(fdf, HostObjective):
            fdf.reraise()
        return r
    finally:
        s.close()
        if own is not None:
            own.close()


class UndefVarError(RuntimeError):
    """What the reference raises when solvesystem's line search finds no step (solve_system.jl:55 reads
    the loop variable `i` outside the loop)."""


def solvesystem(fdf, x_initial: Sequence[float], config: CGConfig, linesearch_config: LinesearchSolveSys,
                reference_throws: bool = False) -> Results:
    """solvesystem(fdf!, x_initial, config, linesearch_config)  (src/engine/solve_system.jl:64-237):
    the Hager–Zhang-type CG of (Yuan 2019) for g(x) = 0, where `fdf!` writes g(x) and returns any
    scalar to trace.  Restated bug for bug — see include/cgo.h, cgo_solver_create_sys.

    When no trial step passes the reference throws UndefVarError instead of returning its
    `:linesearch_failed` record; here that record (status "linesearch_failed", last good iterate) is
    returned, or the exception raised if `reference_throws=True`."""
    if not isinstance(linesearch_config, LinesearchSolveSys):
        raise TypeError("solvesystem takes a LinesearchSolveSys (setupLinesearchSolveSys)")
    s = Solver(fdf, config, linesearch_config)
    try:
        s.set_x0(np.asarray(x_initial, dtype=np.float64))
        s.start()
        while not s.iterate(1 << 40):
            pass
        r = s.results()
    finally:
        s.close()
    if reference_throws and r.status == "linesearch_failed":
        raise UndefVarError("UndefVarError: `i` not defined  (solve_system.jl:55)")
    return r


# ---------------------------------------------------------------------------
# primalbarriermethod! (src/engine/primal_barrier.jl) — a host-side CALLER of minimizeobjectivererun
# ---------------------------------------------------------------------------
@dataclass
class PrimalBarrierConfig:
    """PrimalBarrierConfig{T}  (primal_barrier.jl:130-136)."""
    barrier_tol: float
    barrier_growth_factor: float
    max_iters: int
    t_initial: float
    inf_f0_lb: float


def setupPrimalBarrierConfig(barrier_tol: float, barrier_growth_factor: float, max_iters: int,
                             t_initial: float = math.nan) -> PrimalBarrierConfig:
    """setupPrimalBarrierConfig(barrier_tol, barrier_growth_factor, max_iters; t_initial = NaN)  (:138-154)."""
    return PrimalBarrierConfig(barrier_tol, barrier_growth_factor, int(max_iters), t_initial, 0.0)


@dataclass
class PrimalBarrierResults:
    """PrimalBarrierResults{T,TrT}  (primal_barrier.jl:1-7)."""
    centering_results: List[List[Results]]
    status: str
    iters_ran: int
    t_final: float
    total_objective_evals: int


@dataclass
class BoxConstraints:
    """The box constraints lb ≤ x_d ≤ ub of examples/constrained.jl:18-48 (its `boxhdh!` + the
    `CvxInequalityConstraint` buffers, primal_barrier.jl:38-60) as a device-side descriptor: 2·D strict
    inequalities h(x) < 0, rows 1..D = x − ub, rows D+1..2D = lb − x.  The reference evaluates them as a
    dense 2D×D Jacobian on the host (O(D²) per call); here the log barrier ψ = −Σ log(−h_i) and its
    gradient (primal_barrier.jl:70-94) are element-wise terms inlined into the fused kernels."""
    lb: float
    ub: float

    def n_constraints(self, D: int) -> int:
        return 2 * D


def barrier_objective_source(base: str, box: BoxConstraints) -> str:
    """HIP source of `t·f0(x) + ψ(x)` (evalbarrier!, primal_barrier.jl:111-128) around a base functor.
    `base` is the name of a built-in functor (ObjQuadDiag, ObjRosenPaired, ObjBooth) or the source of a
    `struct BaseObjective { kParam; kPairOnly; eval1; eval2 }`; t lives in the objective's scalar slot."""
    name = base.strip() if base.strip() in ("ObjQuadDiag", "ObjRosenPaired", "ObjBooth") else "BaseObjective"
    pre = "" if name != "BaseObjective" else base
    ub, lb = float(box.ub).hex(), float(box.lb).hex()
    return pre + f"""
struct UserObjective {{
    using B = {name};
    static constexpr bool kParam = B::kParam;
    static constexpr bool kPairOnly = B::kPairOnly;
    __device__ static inline void bar(double x, double &psi, double &dpsi) {{
        const double hu = x - ({ub}), hl = ({lb}) - x;              // fi_evals (examples/constrained.jl:31-32)
        const double cu = hu > 0.0 ? 0.0 : hu, cl = hl > 0.0 ? 0.0 : hl;  // clamp!(fi_evals, -Inf, 0)  (primal_barrier.jl:81)
        psi = -(log(-cu) + log(-cl));                                // ψ = −Σ log(−f_i)  (:82)
        double d = 0.0;
        d -= 1.0 / cu;                                               // dψ[d] −= df_i[d]/f_i  (:85-89), upper row
        d -= -1.0 / cl;                                              //                           lower row
        dpsi = d;
    }}
    __device__ static inline void eval1(double x, double p, double s0, double &f, double &g) {{
        double f0 = 0.0, g0 = 0.0, psi, dpsi;
        B::eval1(x, p, 0.0, f0, g0);
        bar(x, psi, dpsi);
        f += s0 * f0 + psi;                                          // t·f0 + ψ  (:127)
        g = s0 * g0 + dpsi;                                          // df = t·df0 + dψ  (:125)
    }}
    __device__ static inline void eval2(d2 xx, d2 pp, double s0, double &f, d2 &gg) {{
        double f0 = 0.0, psi0, psi1, d0, d1;
        d2 g0;
        B::eval2(xx, pp, 0.0, f0, g0);
        bar(xx.x, psi0, d0);
        bar(xx.y, psi1, d1);
        f += s0 * f0 + (psi0 + psi1);
        gg.x = s0 * g0.x + d0;
        gg.y = s0 * g0.y + d1;
    }}
}};
"""


def _base_objective_source(base: str) -> str:
    name = base.strip() if base.strip() in ("ObjQuadDiag", "ObjRosenPaired", "ObjBooth") else "BaseObjective"
    pre = "" if name != "BaseObjective" else base
    return pre + f"""
struct UserObjective {{
    using B = {name};
    static constexpr bool kParam = B::kParam;
    static constexpr bool kPairOnly = B::kPairOnly;
    __device__ static inline void eval1(double x, double p, double s0, double &f, double &g) {{ B::eval1(x, p, s0, f, g); }}
    __device__ static inline void eval2(d2 xx, d2 pp, double s0, double &f, d2 &gg) {{ B::eval2(xx, pp, s0, f, gg); }}
}};
"""


def primalbarriermethod(constraints: BoxConstraints, f0df0: str, x_initial: Sequence[float],
                        centering_config: CGConfig, linesearch_config: LineSearchConfig,
                        barrier_config: PrimalBarrierConfig, *rerun_config_tuples,
                        param: Optional[np.ndarray] = None, ctx: Optional[Context] = None) -> PrimalBarrierResults:
    """primalbarriermethod!(constraints, f0df0!, hdh!, x_initial, centering_config, linesearch_config,
    barrier_config, rerun_config_tuples...)  (primal_barrier.jl:156-255; algorithm 11.1 of Boyd 2004).

    `(constraints, hdh!)` is a BoxConstraints descriptor and `f0df0` the base objective as device source
    (see barrier_objective_source); every centering step is one minimizeobjectivererun on the GPU.
    As in the reference, `x` is a copy of `x_initial` that is never updated (:172,:214-220): every centering
    step restarts from `x_initial`."""
    x0 = np.ascontiguousarray(x_initial, dtype=np.float64)
    D = x0.size
    bc = barrier_config
    rets: List[List[Results]] = []

    def assemble(status, it, t):                                    # assembleresults!  (:9-35)
        total = sum(int(e) for rr in rets[:it] for r in rr for e in r.trace.objective_evals)
        return PrimalBarrierResults(rets[:it], status, it, t, total)

    if np.any(x0 - constraints.ub >= 0.0) or np.any(constraints.lb - x0 >= 0.0):   # :178-191
        return assemble("infeasible_start", 0, bc.t_initial)
    obj = ElementwiseObjective(D, barrier_objective_source(f0df0, constraints), param=param, ctx=ctx)
    try:
        t = bc.t_initial
        if not math.isfinite(t) or t < 0.0:                         # verifyt0  (:259-276)
            base = ElementwiseObjective(D, _base_objective_source(f0df0), param=param, ctx=ctx)   # f0 alone
            try:
                f_x0 = base(np.empty(D), x0)
            finally:
                base.close()
            t = (f_x0 - bc.inf_f0_lb) * bc.barrier_growth_factor
        for i in range(1, bc.max_iters + 1):                        # :208
            obj.set_scalar(t)
            rets.append(minimizeobjectivererun(obj, x0, centering_config, linesearch_config, *rerun_config_tuples))
            if rets[-1][-1].status != "success":
                return assemble("centering_step_issue", i, t)      # :215-223
            if constraints.n_constraints(D) / t < bc.barrier_tol:   # :229
                return assemble("success", i, t)
            t = bc.barrier_growth_factor * t                        # :240
        return assemble("max_iters_reached", bc.max_iters, t)
    finally:
        obj.close()


def minimizeobjectivererun(fdf, x_initial, config: CGConfig, linesearch_config: LineSearchConfig,
                           *rerun_config_tuples) -> List[Results]:
    """minimizeobjectivererun(fdf!, x_initial, config, ls, rerun_config_tuples...)  (optim.jl:173-208).

    One call of cgo_minimize_rerun: every stage restarts from the previous stage's minimizer, which stays on the GPU
    (device-to-device copy between the stages).  On a sharded context every rank runs the chain on its own shard —
    `x_initial` is the GLOBAL vector, the returned minimizers / gradients are this rank's shards — and all ranks see the
    same statuses, hence the same number of stages."""
    if not isinstance(fdf, DeviceObjective):
        if not callable(fdf):
            raise TypeError("fdf! must be an objective descriptor or a closure fdf(g, x) -> f (no CPU solver path in this package)")
        host = HostObjective(fdf, len(x_initial))   # the reference's own call form: a closure
        try:
            rets = minimizeobjectivererun(host, x_initial, config, linesearch_config, *rerun_config_tuples)
            host.reraise()
            return rets
        finally:
            host.close()
    L = _lib.lib()
    npairs = len(rerun_config_tuples)
    n = fdf.n_local
    x0 = np.ascontiguousarray(fdf.local(np.asarray(x_initial, dtype=np.float64)), dtype=np.float64)
    cfgs = [config] + [t[0] for t in rerun_config_tuples]
    outs = (ResultsC * (1 + npairs))()
    bufs = []
    for i in range(1 + npairs):
        cap = max(cfgs[i].max_iters, 1)
        b = dict(x=np.empty(n), g=np.empty(n), to=np.zeros(cap), tg=np.zeros(cap), ts=np.zeros(cap),
                 te=np.zeros(cap, dtype=np.int64))
        outs[i].minimizer, outs[i].gradient = b["x"].ctypes.data_as(dp), b["g"].ctypes.data_as(dp)
        outs[i].trace_objective, outs[i].trace_grad_norm = b["to"].ctypes.data_as(dp), b["tg"].ctypes.data_as(dp)
        outs[i].trace_step_size = b["ts"].ctypes.data_as(dp)
        outs[i].trace_objective_evals = b["te"].ctypes.data_as(i64p)
        bufs.append(b)
    c0, l0 = config._c(), linesearch_config._c()
    rc = (CGConfigC * max(npairs, 1))(*[t[0]._c() for t in rerun_config_tuples])
    rl = (LSConfigC * max(npairs, 1))(*[t[1]._c() for t in rerun_config_tuples])
    nouts = C.c_int32(0)
    check(L.cgo_minimize_rerun(fdf.ctx._h, fdf._h, x0.ctypes.data_as(dp), C.byref(c0), C.byref(l0),
                               rc, rl, npairs, outs, C.byref(nouts)))
    rets = []
    for i in range(nouts.value):
        b, r = bufs[i], outs[i]
        k = int(r.iters_ran) if isinstance(cfgs[i].trace_status, EnableTrace) else 0
        tr = TraceContainer(b["to"][:k].copy(), b["tg"][:k].copy(), b["ts"][:k].copy(),
                            b["te"][:k].copy(), cfgs[i].trace_status)
        rets.append(Results(r.objective, b["x"], b["g"], int(r.iters_ran),
                            L.cgo_status_name(r.status).decode(), tr, int(r.total_fdf_evals),
                            int(r.total_launches)))
    return rets


# ---------------------------------------------------------------------------
# kernel-level entry points (generic dispatch functions of the reference)
# ---------------------------------------------------------------------------
def updatedir_(u: np.ndarray, df_x: np.ndarray, β: float, ctx: Optional[Context] = None):
    """updatedir!(u, df_x, β)  (cg_flavours.jl:2-15), in place; returns (g·u_new, u_new·u_new)."""
    ctx = ctx or default_context()
    assert u.shape == df_x.shape
    out = np.zeros(2)
    g = np.ascontiguousarray(df_x, dtype=np.float64)
    check(_lib.lib().cgo_kernel_dir(ctx._h, u.ctypes.data_as(dp), g.ctypes.data_as(dp), float(β),
                                    u.size, out.ctypes.data_as(dp)))
    return out[0], out[1]


def getβ(β_config: βConfig, g_next: np.ndarray, g: np.ndarray, u: np.ndarray,
         ctx: Optional[Context] = None) -> float:
    """getβ(β_config, g_next, g, u)::T  (cg_flavours.jl:46-170): one fused pass + scalar work."""
    ctx = ctx or default_context()
    gn = np.ascontiguousarray(g_next, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    b = β_config._c()
    out = C.c_double()
    check(_lib.lib().cgo_getbeta(ctx._h, C.byref(b), gn.ctypes.data_as(dp), g.ctypes.data_as(dp),
                                 u.ctypes.data_as(dp), gn.size, C.byref(out)))
    return out.value


def beta_partials(g_next, g, u, ctx: Optional[Context] = None) -> np.ndarray:
    ctx = ctx or default_context()
    gn = np.ascontiguousarray(g_next, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    out = np.zeros(9)
    check(_lib.lib().cgo_kernel_beta_partials(ctx._h, gn.ctypes.data_as(dp), g.ctypes.data_as(dp),
                                              u.ctypes.data_as(dp), gn.size, out.ctypes.data_as(dp)))
    return out


def evalϕdϕ(fdf: DeviceObjective, a: float, x: np.ndarray, u: np.ndarray):
    """evalϕdϕ!(xp, df_xp, fdf!, a, x, u)  (cg_utils.jl:4-23) → (ϕ, dϕ, df_xp)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    g = np.empty_like(x)
    out = np.zeros(2)
    check(_lib.lib().cgo_kernel_trial(fdf._h, x.ctypes.data_as(dp), u.ctypes.data_as(dp), float(a),
                                      g.ctypes.data_as(dp), out.ctypes.data_as(dp)))
    return out[0], out[1], g


def bench_stream_mix(n: int, reps: int = 9, ctx: Optional[Context] = None):
    """(median_us, best_us) of the accept+dir+trial read/write mix (R x,u,D / W x,u, 40 B/element) without its
    arithmetic, under the engine's pure-HBM streaming policy — the measured ceiling of the dominant launch on this box."""
    ctx = ctx or default_context()
    med, best = C.c_double(), C.c_double()
    check(_lib.lib().cgo_bench_stream_mix(ctx._h, int(n), int(reps), C.byref(med), C.byref(best)))
    return med.value, best.value


def bench_kernel(kind: int, n: int, reps: int = 20, fdf: Optional[DeviceObjective] = None,
                 ctx: Optional[Context] = None):
    ctx = ctx or (fdf.ctx if fdf else default_context())
    ms, b = C.c_double(), C.c_double()
    check(_lib.lib().cgo_bench_kernel(ctx._h, fdf._h if fdf else None, kind, n, reps, C.byref(ms),
                                      C.byref(b)))
    return ms.value, b.value
