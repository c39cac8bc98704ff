"""ctypes binding of the C-ABI library (include/cgo.h → lib/libcgo_hip.so).

The library is the product; this module only declares its signatures.  It
fails loudly when the shared object is missing or cannot be loaded: there is
no Python / CPU fallback for any entry point.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CGO_LIB_PATH") or os.path.join(_PKG, "lib", "libcgo_hip.so")   # CGO_LIB_PATH: A/B builds
CSRC = os.path.join(_PKG, "csrc")

dp = C.POINTER(C.c_double)
i64p = C.POINTER(C.c_int64)


class BetaConfig(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lbfgs_m", C.c_int32), ("mu", C.c_double)]


class CGConfigC(C.Structure):
    _fields_ = [("eps", C.c_double), ("beta", BetaConfig), ("max_iters", C.c_int64),
                ("verbose", C.c_int32), ("trace_enabled", C.c_int32)]


class LSConfigC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("cond_kind", C.c_int32), ("c1", C.c_double),
                ("c2", C.c_double), ("a_max_growth_factor", C.c_double), ("delta1", C.c_double),
                ("max_step_size", C.c_double), ("max_iters", C.c_int64),
                ("zoom_max_iters", C.c_int64), ("feasibility_max_iters", C.c_int64),
                ("discount_factor", C.c_double)]


class LSSConfigC(C.Structure):  # cgo_lss_config — LinesearchSolveSys (solve_system.jl:6-11)
    _fields_ = [("rho", C.c_double), ("sigma", C.c_double), ("s", C.c_double), ("max_iters", C.c_int64)]


class ResultsC(C.Structure):
    _fields_ = [("objective", C.c_double), ("minimizer", dp), ("gradient", dp),
                ("iters_ran", C.c_int64), ("status", C.c_int32), ("_pad", C.c_int32),
                ("trace_objective", dp), ("trace_grad_norm", dp), ("trace_step_size", dp),
                ("trace_objective_evals", i64p), ("total_fdf_evals", C.c_int64),
                ("total_launches", C.c_int64)]


class SolverPolicyC(C.Structure):  # cgo_solver_policy (include/cgo.h)
    _fields_ = [("size", C.c_int32), ("points", C.c_int32), ("resident", C.c_int32), ("controller_depth", C.c_int32),
                ("controller_graph", C.c_int32), ("controller_fused", C.c_int32), ("stored_gradient", C.c_int32),
                ("fused_tail", C.c_int32), ("strict_tail", C.c_int32), ("placement_search", C.c_int32),
                ("placement_stages", C.c_int32), ("placement_max_bytes", C.c_int64), ("lbfgs_form", C.c_int32),
                ("lbfgs_fuse_grad", C.c_int32), ("lbfgs_fuse_trial", C.c_int32), ("lse_fixed_reference", C.c_int32),
                ("resident_points", C.c_int32), ("resident_chunk", C.c_int32), ("hbm_stream_bytes", C.c_double),
                ("reserved", C.c_int32 * 8)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, dp, C.c_int32)
FDF_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, dp, dp, C.c_int64)   # cgo_fdf_fn: f = fdf!(g, x)

# every symbol include/cgo.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_pp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "cgo_version": (C.c_int, []),
    "cgo_build_id": (C.c_char_p, []),
    "cgo_last_error": (C.c_char_p, []),
    "cgo_status_name": (C.c_char_p, [C.c_int32]),
    "cgo_device_count": (C.c_int, [C.POINTER(C.c_int32)]),
    "cgo_check_cg_config": (C.c_int, [C.POINTER(CGConfigC)]),
    "cgo_check_ls_config": (C.c_int, [C.POINTER(LSConfigC)]),
    "cgo_ctx_create": (C.c_int, [C.c_int32, _pp]),
    "cgo_solver_policy_init": (None, [C.POINTER(SolverPolicyC)]),
    "cgo_ctx_set_default_policy": (C.c_int, [_vp, C.POINTER(SolverPolicyC)]),
    "cgo_solver_get_policy": (C.c_int, [_vp, C.POINTER(SolverPolicyC)]),
    "cgo_solver_create_ex": (C.c_int, [_vp, _vp, C.POINTER(CGConfigC), C.POINTER(LSConfigC), C.POINTER(SolverPolicyC), _pp]),
    "cgo_solver_create_sys_ex": (C.c_int, [_vp, _vp, C.POINTER(CGConfigC), C.POINTER(LSSConfigC), C.POINTER(SolverPolicyC), _pp]),
    "cgo_ctx_destroy": (C.c_int, [_vp]),
    "cgo_comm_unique_id": (C.c_int, [_vp]),
    "cgo_ctx_set_comm_rccl": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp]),
    "cgo_ctx_set_comm_callback": (C.c_int, [_vp, C.c_int32, C.c_int32, ALLGATHER_FN, _vp]),
    "cgo_ctx_set_comm_shm": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_char_p, C.c_int32]),
    "cgo_shm_unlink": (C.c_int, [C.c_char_p]),
    "cgo_ctx_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cgo_ctx_exchange_stats": (C.c_int, [_vp, i64p, dp, dp, C.c_int32]),
    "cgo_ctx_comm_connect_devices": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "cgo_rccl_available": (C.c_int, []),
    "cgo_objective_create": (C.c_int, [_vp, C.c_int32, C.c_int64, C.c_int64, C.c_int64, _pp]),
    "cgo_objective_create_from_source": (C.c_int, [_vp, C.c_char_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, _pp]),
    "cgo_objective_create_callback": (C.c_int, [_vp, FDF_FN, _vp, C.c_int64, C.c_int64, C.c_int64, _pp]),
    "cgo_objective_destroy": (C.c_int, [_vp]),
    "cgo_objective_set_param_host": (C.c_int, [_vp, C.c_int32, dp]),
    "cgo_objective_fill_param": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_uint64, C.c_double, C.c_double]),
    "cgo_objective_set_scalar": (C.c_int, [_vp, C.c_int32, C.c_double]),
    "cgo_objective_set_cost_class": (C.c_int, [_vp, C.c_int32]),
    "cgo_objective_eval_host": (C.c_int, [_vp, dp, dp, dp]),
    "cgo_solver_create": (C.c_int, [_vp, _vp, C.POINTER(CGConfigC), C.POINTER(LSConfigC), _pp]),
    "cgo_solver_destroy": (C.c_int, [_vp]),
    "cgo_solver_set_x0_host": (C.c_int, [_vp, dp]),
    "cgo_solver_set_x0_fill": (C.c_int, [_vp, C.c_int32, C.c_uint64, C.c_double, C.c_double]),
    "cgo_solver_set_x0_device": (C.c_int, [_vp, _vp]),
    "cgo_solver_results_device": (C.c_int, [_vp, _vp, _vp]),
    "cgo_solver_start": (C.c_int, [_vp]),
    "cgo_solver_iterate": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_int32)]),
    "cgo_solver_results": (C.c_int, [_vp, C.POINTER(ResultsC)]),
    "cgo_solver_trial_log": (C.c_int, [_vp, C.c_int64, dp, dp, dp, i64p]),
    "cgo_solver_profile_enable": (C.c_int, [_vp, C.c_int32]),
    "cgo_solver_profile_reset": (C.c_int, [_vp]),
    "cgo_solver_profile_get": (C.c_int, [_vp, C.c_int32, i64p, dp, dp]),
    "cgo_kernel_kind_name": (C.c_char_p, [C.c_int32]),
    "cgo_solver_kernel_family": (C.c_char_p, [_vp]),
    "cgo_solver_controller_launches": (C.c_int64, [_vp]),
    "cgo_solver_resident_stats": (C.c_int, [_vp, i64p, i64p, i64p]),
    "cgo_solver_lbfgs_stats": (C.c_int, [_vp, i64p, i64p, i64p]),
    "cgo_num_kernel_kinds": (C.c_int, []),
    "cgo_solver_kernel_symbol": (C.c_int, [_vp, C.c_int32, C.c_char_p, C.c_int32]),
    "cgo_evalwolfeconditions": (C.c_int, [C.POINTER(LSConfigC), C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cgo_evalbacktrackcondition": (C.c_int, [C.POINTER(LSConfigC), C.c_double, C.c_double, C.c_double, C.c_double,
                                             C.POINTER(C.c_int32)]),
    "cgo_check_lss_config": (C.c_int, [C.POINTER(LSSConfigC)]),
    "cgo_lss_default_max_iters": (C.c_int64, [C.c_double]),
    "cgo_solver_create_sys": (C.c_int, [_vp, _vp, C.POINTER(CGConfigC), C.POINTER(LSSConfigC), _pp]),
    "cgo_solvesystem": (C.c_int, [_vp, _vp, dp, C.POINTER(CGConfigC), C.POINTER(LSSConfigC), C.POINTER(ResultsC)]),
    "cgo_minimize": (C.c_int, [_vp, _vp, dp, C.POINTER(CGConfigC), C.POINTER(LSConfigC), C.POINTER(ResultsC)]),
    "cgo_minimize_rerun": (C.c_int, [_vp, _vp, dp, C.POINTER(CGConfigC), C.POINTER(LSConfigC),
                                     C.POINTER(CGConfigC), C.POINTER(LSConfigC), C.c_int32,
                                     C.POINTER(ResultsC), C.POINTER(C.c_int32)]),
    "cgo_kernel_dir": (C.c_int, [_vp, dp, dp, C.c_double, C.c_int64, dp]),
    "cgo_kernel_beta_partials": (C.c_int, [_vp, dp, dp, dp, C.c_int64, dp]),
    "cgo_getbeta": (C.c_int, [_vp, C.POINTER(BetaConfig), dp, dp, dp, C.c_int64, dp]),
    "cgo_kernel_trial": (C.c_int, [_vp, dp, dp, C.c_double, dp, dp]),
    "cgo_bench_kernel": (C.c_int, [_vp, _vp, C.c_int32, C.c_int64, C.c_int32, dp, dp]),
    "cgo_bench_stream_mix": (C.c_int, [_vp, C.c_int64, C.c_int32, dp, dp]),
    "cgo_solver_placement_info": (C.c_int, [_vp, dp, dp, C.POINTER(C.c_int32)]),
}


class CgoError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[cgo error {code}] {msg}")
        self.code = code
        self.msg = msg


def build(force: bool = False) -> str:
    """Compile libcgo_hip.so for gfx950 with hipcc (recipe: csrc/Makefile)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean", "-s"], check=True)
    subprocess.run(["make", "-C", CSRC, "-s", "-j4"], check=True)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce " + LIB_PATH)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load libcgo_hip.so.  Raises (never falls back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC}` "
                "(or __graft_entry__.build()). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError = ABI drift: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().cgo_last_error()
        text = msg.decode("utf-8", "replace") if msg else ""
        if rc == 1 and text.startswith("AssertionError"):
            raise AssertionError(text[len("AssertionError: "):])
        raise CgoError(rc, text)
