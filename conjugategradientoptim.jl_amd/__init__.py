"""conjugategradientoptim.jl_amd — MI355X-native inner-iteration engine for
ConjugateGradientOptim.jl's nonlinear-CG / quasi-Newton hot path.

  csrc/   hand-written HIP kernels (gfx950) + host engine + the C ABI (include/cgo.h)
  lib/    libcgo_hip.so (built in-tree by csrc/Makefile)
  api.py  host-side mirror of the reference's interface over that C ABI
  julia/  the same binding for Julia hosts (ccall)

The directory name carries a dot, so import it through the repo-root alias:
    import cgo_amd as cgo
"""
from . import _lib
from ._lib import CgoError, build, lib
from .api import *  # noqa: F401,F403
from . import api

__all__ = [n for n in dir(api) if not n.startswith("_")] + ["CgoError", "build", "lib"]
