"""CPU tier: the C-ABI library loads, exports every symbol include/cgo.h declares,
agrees with the header on struct layout, enforces the reference's config
@asserts, and FAILS LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cgo.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cgo_[a-z0-9_]+)\s*\(", src)) - {"cgo_allgather_fn"})


def test_library_exports_every_declared_symbol(cgo):
    from cgo_amd import _lib
    L = _lib.lib()
    names = header_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/cgo.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (cgo_[a-z0-9_]+)", out))
    assert set(names) <= exported
    assert L.cgo_version() == 100


def test_struct_layout_matches_header(cgo, tmp_path):
    """Compile a tiny C program against include/cgo.h and compare sizeof/offsetof with ctypes."""
    from cgo_amd import _lib
    prog = tmp_path / "layout.c"
    prog.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "cgo.h"
int main(void){
 printf("%zu %zu %zu %zu %zu\\n", sizeof(cgo_beta_config), sizeof(cgo_cg_config), sizeof(cgo_ls_config), sizeof(cgo_results), sizeof(cgo_lss_config));
 printf("%zu %zu %zu %zu\\n", offsetof(cgo_cg_config, max_iters), offsetof(cgo_ls_config, max_iters), offsetof(cgo_results, trace_objective), offsetof(cgo_lss_config, max_iters));
 printf("%d %d %d\\n", CGO_NUM_STATUS, CGO_BETA_LBFGS, CGO_OBJ_LSE);
 return 0; }''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    l1, l2, l3 = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert [int(v) for v in l1.split()] == [C.sizeof(_lib.BetaConfig), C.sizeof(_lib.CGConfigC),
                                            C.sizeof(_lib.LSConfigC), C.sizeof(_lib.ResultsC), C.sizeof(_lib.LSSConfigC)]
    assert [int(v) for v in l2.split()] == [_lib.CGConfigC.max_iters.offset, _lib.LSConfigC.max_iters.offset,
                                            _lib.ResultsC.trace_objective.offset, _lib.LSSConfigC.max_iters.offset]
    assert [int(v) for v in l3.split()] == [20, 7, 3]


def test_status_names_are_the_reference_symbols(cgo):
    from cgo_amd import _lib
    from oracle import oracle as O
    L = _lib.lib()
    for i, name in enumerate(O.STATUS_NAMES):
        assert L.cgo_status_name(i).decode() == name
    assert L.cgo_status_name(99).decode() == "unknown"
    kinds = [L.cgo_kernel_kind_name(k).decode() for k in range(L.cgo_num_kernel_kinds())]
    assert kinds[:3] == ["init", "trial", "accept_dir_trial"]


def test_config_asserts_mirror_reference(cgo):
    with pytest.raises(AssertionError, match="ϵ"):      # types.jl:187
        cgo.setupCGConfig(1.0, cgo.HagerZhang(), cgo.EnableTrace())
    with pytest.raises(AssertionError):                  # types.jl:187
        cgo.setupCGConfig(0.0, cgo.HagerZhang(), cgo.EnableTrace())
    with pytest.raises(AssertionError, match="c1 < c2"):  # nocedal.jl:22
        cgo.setupStrongWolfeBisection(0.9, 0.8)
    with pytest.raises(AssertionError, match="growth"):   # nocedal.jl:26
        cgo.setupStrongWolfeBisection(1e-5, 0.8, a_max_growth_factor=1.0)
    cfg = cgo.setupCGConfig(1e-5, cgo.YuanWangSheng(0.1), cgo.DisableTrace(), max_iters=7)
    assert cfg.max_iters == 7 and cfg.β_config.μ == 0.1 and cfg.verbose is False
    ls = cgo.setupStrongWolfeBisection(1e-5, 0.8)       # defaults of nocedal.jl:17-19
    assert (ls.a_max_growth_factor, ls.max_iters, ls.zoom_max_iters) == (2.0, 1000, 100)


def test_no_cpu_fallback(cgo):
    """Without a GPU the product refuses to run — also for a host closure, whose solve is the GPU engine's
    (cgo_objective_create_callback), not a CPU one."""
    from cgo_amd import _lib
    cnt = C.c_int32(-1)
    assert _lib.lib().cgo_device_count(C.byref(cnt)) == 0
    if cnt.value == 0:
        with pytest.raises(cgo.CgoError) as e:
            cgo.Context(0)
        assert e.value.code == 4 and "no CPU fallback" in e.value.msg   # CGO_ENODEV
    cfg = cgo.setupCGConfig(1e-5, cgo.HagerZhang(), cgo.EnableTrace())
    if cnt.value == 0:
        with pytest.raises(cgo.CgoError) as e:
            cgo.minimizeobjective(lambda g, x: 0.0, np.zeros(2), cfg, cgo.setupStrongWolfeBisection(1e-5, 0.8))
        assert e.value.code == 4
    with pytest.raises(TypeError, match="no CPU solver path"):
        cgo.Solver("not an objective", cfg, cgo.setupStrongWolfeBisection(1e-5, 0.8))


def test_product_does_not_link_the_oracle(cgo):
    from cgo_amd import _lib
    out = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in out and "sim_" not in out
    ldd = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "hostsim" not in ldd and "libamdhip64" in ldd


def test_shard_extents(cgo):
    for n in (2, 3, 10, 11, 1000, 100003, 10**8):
        for w in (1, 2, 3, 4, 8):
            ext = [cgo.shard_extent(n, r, w) for r in range(w)]
            assert ext[0][0] == 0 and sum(e[1] for e in ext) == n
            for r in range(1, w):
                assert ext[r][0] == ext[r - 1][0] + ext[r - 1][1]
                assert ext[r][0] % 2 == 0      # pairs never straddle a shard boundary


def _build_c_example(tmp_path):
    exe = tmp_path / "booth_min_c"
    libdir = os.path.join(ROOT, "conjugategradientoptim.jl_amd", "lib")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "booth_min.c"), "-L", libdir, "-lcgo_hip",
                    f"-Wl,-rpath,{libdir}", "-lm", "-o", str(exe)], check=True)
    return exe


def test_plain_c_client_links_and_fails_loudly_without_gpu(cgo, tmp_path):
    """include/cgo.h is plain C11 and the library links from a C program (what any FFI binds).
    Without a gfx950 device the client must stop at cgo_ctx_create with CGO_ENODEV — never compute on the CPU."""
    exe = _build_c_example(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    if r.returncode != 0:
        assert r.returncode == 1 and "cgo_ctx_create" in r.stderr and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_plain_c_client_runs_examples_min_jl(cgo, gpu_ctx, tmp_path):
    exe = _build_c_example(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "status success" in r.stdout


def test_condition_evaluators_kats(cgo):
    """evalwolfeconditions (wolfe.jl:219-294) and evalbacktrackcondition (geometric.jl:164-186) through the ABI —
    scalar host functions, the same code the engine and the on-device controller run — against the hand-derived
    values the oracle is pinned with (tests/test_oracle.py::test_wolfe_condition_kats)."""
    import numpy as np
    u = np.array([3.0, 4.0])                      # u·u = 25
    w = cgo.Wolfe(0.25, 0.5)                      # ϕ0 = 10, dϕ0 = −8, a = 0.5: RHS1 = 9, RHS2 = −4
    assert cgo.evalwolfeconditions(w, 9.0, -4.0, 0.5, u, 10.0, -8.0) == (True, True)
    assert cgo.evalwolfeconditions(w, 9.0000001, -4.0, 0.5, u, 10.0, -8.0) == (False, True)
    assert cgo.evalwolfeconditions(w, 9.0, -4.0000001, 0.5, u, 10.0, -8.0) == (True, False)
    y = cgo.YuanWeiLuWolfe(0.25, 0.5, 0.125)      # min(1, 1.5625) = 1 → RHS1 = 9.5 ; min(1, 3.125) = 1 → RHS2 = −3
    assert cgo.evalwolfeconditions(y, 9.5, -3.0, 0.5, u, 10.0, -8.0) == (True, True)
    assert cgo.evalwolfeconditions(y, 9.5000001, -3.0, 0.5, 25.0, 10.0, -8.0) == (False, True)
    assert cgo.evalwolfeconditions(y, 9.5, -3.0000001, 0.5, u, 10.0, -8.0) == (True, False)
    a = cgo.Armijo(0.25)                          # ϕ0 − ϕa ≥ −c1·a·dϕ0 = 1
    assert cgo.evalbacktrackcondition(a, 9.0, 0.5, 10.0, -8.0) is True
    assert cgo.evalbacktrackcondition(a, 9.0000001, 0.5, 10.0, -8.0) is False
    assert cgo.evalbacktrackcondition(a, float("inf"), 0.5, 10.0, -8.0) is False      # non-finite ⇒ false (:175-177)
    with pytest.raises(AssertionError):
        cgo.evalwolfeconditions(cgo.Wolfe(0.5, 0.25), 9.0, -4.0, 0.5, u, 10.0, -8.0)   # wolfe.jl:278
