"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/bazinga_hip.h declares, the ctypes mirrors have the C layout, and — with no GPU in
this container — compute entry points fail loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bazinga_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bz_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(bz):
    lib = bz._lib.load()
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in bazinga_hip.h but not exported"
    assert set(names) == set(bz._lib.SIGNATURES), "ctypes table and header disagree"


def test_struct_layouts_match_the_header(bz):
    """Compile a C program against the header and compare sizeof/offsetof with the ctypes mirrors."""
    L = bz._lib
    structs = {"bz_ctx_opts": L.CtxOpts, "bz_problem_desc": L.ProblemDesc, "bz_panoc_opts": L.PanocOpts,
               "bz_panoc_stats": L.PanocStats, "bz_alps_opts": L.AlpsOpts, "bz_alps_stats": L.AlpsStats,
               "bz_profile_rec": L.ProfileRec}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, st in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.c")
        open(src, "w").write("\n".join(lines))
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", src, "-o", exe])
        out = subprocess.check_output([exe], text=True)
    got = dict(l.split() for l in out.strip().splitlines())
    for cname, st in structs.items():
        assert int(got[cname]) == C.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(st, fname).offset, f"{cname}.{fname}"


def test_header_is_plain_c(bz):
    """The boundary is a C ABI: the header must compile as C99 and mention no torch/C++ types."""
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.c")
        open(src, "w").write(f'#include "{HEADER}"\nint main(void){{return 0;}}\n')
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-c", src, "-o", os.path.join(td, "t.o")])
    text = open(HEADER).read()
    assert "torch" not in text and "std::" not in text


def test_defaults_mirror_the_reference(bz):
    L = bz._lib
    o = L.PanocOpts()
    L.load().bz_panoc_default_opts(C.byref(o))
    assert (o.tol, o.maxit, o.freq, o.minimum_gamma, o.alpha, o.beta, o.max_backtracks, o.lbfgs_memory) == \
        (1e-8, 1000, 10, 1e-7, 0.95, 0.5, 20, 5)
    # upstream: Lf = nothing, gamma = nothing, adaptive = (gamma === nothing)
    assert (o.gamma, o.Lf, o.adaptive) == (0.0, 0.0, -1)
    a = L.AlpsOpts()
    L.load().bz_alps_default_opts(C.byref(a), L.BZ_F64)
    # src/algorithms/alps.jl:14-25
    assert a.tol_prim == 1e-6 and a.tol_dual == 1e-6 and abs(a.inner_tol - 1e-2) < 1e-15
    assert (a.maxit, a.theta_penalty, a.kappa_penalty, a.kappa_tol, a.subsolver_maxit) == (100, 0.8, 0.5, 0.1, 10 ** 9)
    assert a.warm_start == 0          # alps.jl:64 as written unless the caller opts in


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="checks behaviour on a GPU-less host")
def test_no_silent_cpu_fallback(bz):
    """Without a device the product path must raise, never compute on the CPU."""
    with pytest.raises(bz.BazingaHipError) as ei:
        bz.Context()
    assert ei.value.code == bz._lib.BZ_ERR_HIP
    n = 8
    f, g, c, D = bz.DiagQuadratic(np.ones(n), np.ones(n)), bz.NormL1(1.0), bz.IdentityFunction(), bz.FreeSet()
    with pytest.raises(bz.BazingaHipError):
        bz.alps(f, g, c, D, np.zeros(n), np.zeros(n))
    with pytest.raises(bz.BazingaHipError):
        bz.alps(f, g, c, D, np.zeros(n), np.zeros(n), resident=False)


def test_product_does_not_import_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "bazinga.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn
                assert "libbz_oracle" not in text, fn
    code = "import sys; sys.path.insert(0, %r); import bazinga_jl_amd; " \
           "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)" % ROOT
    subprocess.check_call([sys.executable, "-c", code])


def test_julia_shim_structs_mirror_the_ctypes_twin(bz):
    """julia/BazingaHIP.jl cannot be executed here (no Julia): at least its struct mirrors are compared, field by field —
    name, order and width — with the ctypes structs that test_struct_layouts_match_the_header pins to the C header, and
    the defaults its keyword constructors give PanocOpts / AlpsOpts with what bz_*_default_opts fill in."""
    import re
    L = bz._lib
    src = open(os.path.join(ROOT, "julia", "BazingaHIP.jl")).read()
    width = {"Int32": 4, "Int64": 8, "Float64": 8, "Ptr{Cvoid}": 8, "UInt64": 8}
    ctw = {C.c_int32: 4, C.c_int64: 8, C.c_double: 8, C.c_void_p: 8}
    twins = {"CtxOpts": L.CtxOpts, "ProblemDesc": L.ProblemDesc, "PanocOpts": L.PanocOpts, "PanocStats": L.PanocStats,
             "AlpsOpts": L.AlpsOpts, "AlpsStats": L.AlpsStats}
    parsed = {}
    for name, ct in twins.items():
        m = re.search(r"(?:mutable )?struct " + name + r"\n(.*?)\nend", src, re.S)
        assert m, f"struct {name} not found in the Julia shim"
        body = re.sub(r"#.*", "", m.group(1))
        fields = re.findall(r"(\w+)::([\w{}]+)(?:\s*=\s*([^;\n]+))?", body)
        parsed[name] = fields
        jl = [(f, width[t]) for f, t, _ in fields]
        cf = []
        for fname, ftype in ct._fields_:
            cf.append((fname, ctw[ftype] if ftype in ctw else C.sizeof(ftype)))
        assert jl == cf, f"{name}: Julia {jl} vs ctypes {cf}"
    o = L.PanocOpts()
    L.load().bz_panoc_default_opts(C.byref(o))
    for f, t, dflt in parsed["PanocOpts"]:
        assert dflt is not None and float(eval(dflt.strip().replace("_", ""))) == float(getattr(o, f)), f"PanocOpts.{f}: {dflt}"
    a = L.AlpsOpts()
    L.load().bz_alps_default_opts(C.byref(a), L.BZ_F64)
    for f, t, dflt in parsed["AlpsOpts"]:
        want = float(getattr(a, f))
        got = float(eval(dflt.strip().replace("_", "").replace("cbrt(1e-6)", "1e-6 ** (1 / 3)")))
        assert abs(got - want) <= 1e-15 * max(1.0, abs(want)), f"AlpsOpts.{f}: {dflt} vs {want}"


def test_julia_shim_ccalls_match_the_exported_signatures(bz):
    """... and every `ccall` of the shim names an exported entry point with as many arguments as the ctypes twin passes,
    pointers where it passes pointers."""
    import re
    L = bz._lib
    src = open(os.path.join(ROOT, "julia", "BazingaHIP.jl")).read()
    src = re.sub(r"#.*", "", src)
    calls = re.findall(r"ccall\(\(:(\w+), lib\),\s*(\w+),\s*\((.*?)\)\s*[,)]", src, re.S)
    assert len(calls) >= 7
    lib = L.load()
    for name, ret, types in calls:
        assert name in L.SIGNATURES, f"{name} is not an entry point of the library"
        assert hasattr(lib, name)
        res, args = L.SIGNATURES[name]
        depth, parts, cur = 0, [], ""
        for ch in types:                                   # split on top-level commas (Ref{Ptr{Cvoid}} holds none, but be safe)
            if ch == "{":
                depth += 1
            elif ch == "}":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur.strip())
        assert len(parts) == len(args), f"{name}: Julia passes {parts}, the library takes {len(args)} arguments"
        for jt, ct in zip(parts, args):
            is_ptr_j = jt.startswith(("Ptr{", "Ref{")) or jt == "Cstring"
            is_ptr_c = ct in (C.c_void_p, C.c_char_p) or hasattr(ct, "contents")
            assert is_ptr_j == is_ptr_c, f"{name}: {jt} vs {ct}"
            if not is_ptr_j:
                assert {"Cint": 4, "Int32": 4, "Int64": 8, "Cdouble": 8, "Float64": 8}[jt] == C.sizeof(ct), f"{name}: {jt} vs {ct}"
        assert (ret == "Cvoid") == (res is None) and (ret == "Cstring") == (res is C.c_char_p)
