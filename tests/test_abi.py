"""CPU: the C-ABI library loads and exports exactly the symbols include/juqbox_hip.h declares; the
argument validation that precedes any device work returns the documented error codes."""
import ctypes
import os
import re

import numpy as np
import pytest
from conftest import ROOT, case_inputs


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "juqbox_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(jq_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from juqbox_jl_amd import _lib
    L = _lib.load()
    declared = header_symbols()
    assert declared, "no declarations parsed"
    assert sorted(_lib.SYMBOLS) == declared, "python binding table and header disagree"
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.jq_version()


def _problem(jq, params, **over):
    from juqbox_jl_amd import _lib
    f = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))
    keep = [f(params.Hconst), np.concatenate([f(h) for h in params.Hsym_ops]),
            np.concatenate([f(h) for h in params.Hanti_ops]), f(params.Uinit), f(params.Utarget_r),
            f(params.Utarget_i), f(params.wmat_real), f(params.Cfreq)]
    vals = dict(Ntot=params.Ntot, N=params.N, Ncoupled=params.Ncoupled, Nfreq=params.Nfreq, nsteps=params.nsteps,
                neumann_terms=params.linear_solver.max_iter, objFuncType=params.objFuncType, reserved=0, T=params.T)
    vals.update(over)
    prob = _lib.jq_problem(vals["Ntot"], vals["N"], vals["Ncoupled"], vals["Nfreq"], vals["nsteps"],
                           vals["neumann_terms"], vals["objFuncType"], vals["reserved"], vals["T"],
                           *[a.ctypes.data_as(_lib.c_dp) for a in keep])
    return prob, keep


@pytest.mark.parametrize("over,code", [
    (dict(nsteps=0), -1), (dict(T=0.0), -1), (dict(N=0), -1), (dict(reserved=7), -1), (dict(objFuncType=9), -1),
    (dict(neumann_terms=-1), -1), (dict(Ntot=97), -3), (dict(Ncoupled=0), -3), (dict(Ncoupled=5), -3),
])
def test_create_validates_before_touching_the_device(jq, over, code):
    from juqbox_jl_amd import _lib
    L = _lib.load()
    params, info, pcof, _ = case_inputs("swap02")
    prob, keep = _problem(jq, params, **over)
    h = ctypes.c_void_p()
    rc = L.jq_create(ctypes.byref(prob), ctypes.byref(h))
    assert rc == code
    assert h.value is None
    assert len(L.jq_last_error(None)) > 0


def test_null_arguments_are_rejected():
    from juqbox_jl_amd import _lib
    L = _lib.load()
    assert L.jq_create(None, None) == _lib.JQ_EINVAL
    h = ctypes.c_void_p()
    assert L.jq_create(None, ctypes.byref(h)) == _lib.JQ_EINVAL
    assert L.jq_traceobjgrad(None, None, 0, 0, None, None, None, None) == _lib.JQ_EINVAL
    assert L.jq_last_timing(None, None) == _lib.JQ_EINVAL
    L.jq_destroy(None)   # no-op


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under juqbox.jl_amd/ may import, load or link it."""
    pkg = os.path.join(ROOT, "juqbox.jl_amd")
    bad = re.compile(r"(^\s*(import|from)\s+oracle\b)|libjuqbox_oracle|\bjqo_|oracle/|oracle\.oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert not bad.search(txt), fn
