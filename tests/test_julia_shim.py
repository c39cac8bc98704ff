"""CPU tier: static checks of julia/ConjugateGradientOptimAMD.jl — there is no `julia` in this image, so no interpreter has
ever parsed the shim (VERDICT r03 missing #6).  What CAN be checked without one: block structure, that every `ccall` names a
symbol the library exports with the number of arguments include/cgo.h declares, and that the structs passed by reference
list the fields of their C twins in order."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "conjugategradientoptim.jl_amd", "julia", "ConjugateGradientOptimAMD.jl")
HDR = os.path.join(ROOT, "include", "cgo.h")
LIB = os.path.join(ROOT, "conjugategradientoptim.jl_amd", "lib", "libcgo_hip.so")


def _strip(src):
    """Julia source without comments, strings and character literals (their contents replaced by blanks)."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("#=", i):
            j = src.index("=#", i) + 2
            out.append(" " * (j - i)); i = j
        elif c == "#":
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif src.startswith('"""', i):
            j = src.index('"""', i + 3) + 3
            out.append('""' + "".join("\n" if ch == "\n" else " " for ch in src[i + 2:j - 1]) + '"'); i = j
        elif c == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('"' + " " * (j - i - 1) + '"'); i = j + 1
        elif c == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and src[i + 3] == "'")):   # character literal, not a transpose
            j = i + (3 if src[i + 2] == "'" else 4)
            out.append(" " * (j - i)); i = j
        else:
            out.append(c); i += 1
    return "".join(out)


def _split_args(s):
    """top-level comma split of the inside of a parenthesis"""
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip()); cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        parts.append("".join(cur).strip())
    return parts


def _balanced(s, i):
    """index just past the parenthesis that opens at s[i]"""
    depth = 0
    for j in range(i, len(s)):
        if s[j] in "([{":
            depth += 1
        elif s[j] in ")]}":
            depth -= 1
            if depth == 0:
                return j + 1
    raise AssertionError("unbalanced parenthesis")


def test_blocks_and_brackets_balance():
    """One stack for brackets and blocks: every `end` closes the innermost open block (or is an index inside `[…]`), every
    bracket closes its own kind, nothing is left open at the end of the file."""
    src = _strip(open(JL, encoding="utf-8").read())
    openers = {"function", "struct", "if", "for", "while", "begin", "let", "do", "module", "try", "macro", "quote", "abstract", "primitive"}
    stack = []
    for m in re.finditer(r"[()\[\]{}]|(?<![\w@.:])(?:mutable\s+struct|[a-z]+)(?![\w!])", src):
        t = m.group(0)
        line = src.count("\n", 0, m.start()) + 1
        if t in "([{":
            stack.append(t)
        elif t in ")]}":
            assert stack and {"(": ")", "[": "]", "{": "}"}.get(stack.pop()) == t, f"bracket mismatch at line {line}"
        elif t == "end":
            if stack and stack[-1] == "[":
                continue                                              # a[end]
            assert stack and stack[-1] == "block", f"`end` without an open block at line {line}"
            stack.pop()
        elif t in openers or t.startswith("mutable"):
            if t in ("for", "if", "while") and stack and stack[-1] in "([{":
                continue                                              # generator / filter inside brackets: no block
            if t == "struct" and stack and stack[-1] == "mutable":
                continue
            stack.append("block")
    assert not stack, stack[-5:]


def _c_prototypes():
    hdr = re.sub(r"/\*.*?\*/", " ", open(HDR, encoding="utf-8").read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(cgo_\w+)\s*\(([^;{]*?)\)\s*;", hdr):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(_split_args(args))
    return protos


def test_every_ccall_names_an_exported_symbol_with_the_declared_number_of_arguments():
    src = _strip(open(JL, encoding="utf-8").read())
    protos = _c_prototypes()
    exported = set()
    if os.path.exists(LIB):
        nm = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True).stdout
        exported = {ln.split()[-1] for ln in nm.splitlines() if ln.strip()}
    seen = 0
    for m in re.finditer(r"\bccall\s*\(", src):
        end = _balanced(src, m.end() - 1)
        parts = _split_args(src[m.end():end - 1])
        sym = re.match(r"\(\s*:(\w+)\s*,\s*libcgo\s*\)", parts[0])
        assert sym, parts[0]
        name = sym.group(1)
        assert name in protos, f"{name} is not declared in include/cgo.h"
        if exported:
            assert name in exported, f"{name} is not exported by libcgo_hip.so"
        types = parts[2]
        assert types.startswith("(") and types.endswith(")"), (name, types)
        ntypes = len(_split_args(types[1:-1]))
        nargs = len(parts) - 3
        assert ntypes == nargs, f"ccall {name}: {ntypes} argument types, {nargs} arguments"
        assert ntypes == protos[name], f"ccall {name}: {ntypes} arguments, include/cgo.h declares {protos[name]}"
        seen += 1
    assert seen >= 25


def _jl_struct_fields(src, name):
    m = re.search(r"struct\s+" + name + r"\b(.*?)\n\s*end\b", src, flags=re.S)
    assert m, name
    return [f.group(1) for f in re.finditer(r"^\s*(\w+)\s*::", m.group(1), flags=re.M)]


def _c_struct_fields(name):
    hdr = re.sub(r"/\*.*?\*/", " ", open(HDR, encoding="utf-8").read(), flags=re.S)
    m = re.search(r"typedef\s+struct\s+" + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", hdr, flags=re.S)
    assert m, name
    out = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if decl:
            for piece in decl.split(","):
                out.append(re.sub(r"\[.*\]", "", piece.strip().split()[-1]).lstrip("*"))
    return out


@pytest.mark.parametrize("jl,c", [("SolverPolicy", "cgo_solver_policy"), ("CCGConfig", "cgo_cg_config"), ("CLSConfig", "cgo_ls_config")])
def test_structs_passed_by_reference_list_the_c_fields_in_order(jl, c):
    src = _strip(open(JL, encoding="utf-8").read())
    jf, cf = _jl_struct_fields(src, jl), _c_struct_fields(c)
    assert len(jf) == len(cf), (jf, cf)
    if jl == "SolverPolicy":                      # same names there; the config structs flatten / rename nested members
        assert jf == cf, (jf, cf)


def test_the_shim_defines_every_name_the_reference_examples_call_on_the_module():
    """The names `examples/min.jl`, `examples/constrained.jl` and `test/runtests.jl` qualify with the module name, plus the
    module's export list (src/ConjugateGradientOptim.jl:23-29).  `setupCvxInequalityConstraint` (the dense-Jacobian constraint
    container of examples/constrained.jl:52) is the one deliberate absentee: general constraints are out of scope, the box
    constraints of that example are a `BoxConstraints` descriptor here."""
    src = _strip(open(JL, encoding="utf-8").read())
    names = ["setupCGConfig", "EnableTrace", "DisableTrace", "LiuStorrey", "setupStrongWolfeBisection", "setupBroydenFamily", "YuanWangSheng",
             "WolfeBisection", "SallehAlhawarat", "HagerZhang", "setupPrimalBarrierConfig", "primalbarriermethod!", "minimizeobjective",
             "YuanWeiLuWolfe", "Wolfe", "Backtracking", "Armijo", "TraceContainer", "Results", "LineSearchContainer", "solvesystem",
             "minimizeobjectivererun", "setupLinesearchSolveSys", "StrongWolfeBisection"]
    for n in names:
        pat = r"(?:function\s+|struct\s+|^\s*)" + re.escape(n) + r"(?:\(|\{|\s|$|::)"
        assert re.search(pat, src, flags=re.M), f"{n} is not defined in the shim"
    exports = re.search(r"^export\s+(.*)$", src, flags=re.M).group(1)
    for n in ("TraceContainer", "EnableTrace", "DisableTrace", "Results", "LineSearchContainer", "solvesystem", "minimizeobjective"):
        assert re.search(r"\b" + n + r"\b", exports), f"{n} is not exported"
