"""GPU (-m gpu): full / complex leakage weights (`use_custom_forbidden`, src/evalobjgrad.jl:214-232) through jq_update_wmat, and the
sparse (CSC) operator entry, against the CPU oracle.

PARITY-UNPINNED IN THE REFERENCE (no test, example or golden uses the branch): the oracle restates :700, :716-718, :862, :882-888,
:2183-2228 and is pinned by tests/test_dense_wmat.py (Diagonal equivalence, state history, finite differences); here the device must
agree with it at the reference's own tolerance 1e-10 on every kernel family that carries the low-rank terms, and refuse -- never
silently evaluate other weights -- everywhere else."""
import os

import numpy as np
import pytest

from conftest import case_inputs
from test_dense_wmat import forbidden_problem
from test_gpu_random import random_problem

pytestmark = pytest.mark.gpu

TOL = 1e-10
def make_wa(jq, p, ncoef, env=None, **kw):
    """env: options of the new handle in the historic spelling ({"JQ_QUAD": "0"} = option quad=0)"""
    return jq.Working_Arrays_HIP(p, ncoef, options=env, **kw)


def set_forbidden(p, rng, nforb, complex_states=True):
    fs = rng.standard_normal((p.Ntot, nforb)) + (1j * rng.standard_normal((p.Ntot, nforb)) if complex_states else 0)
    fs = fs / np.linalg.norm(fs, axis=0)
    fw = 0.5 + rng.random(nforb)
    W = sum(fw[k] * np.outer(fs[:, k], np.conj(fs[:, k])) for k in range(nforb))
    p.wmat_real = np.asfortranarray(W.real.copy())
    p.wmat_imag = np.asfortranarray(W.imag.copy())


def compare(jq, p, pcof, wa, family=None, ensembles=(), rng=None, tol=TOL):
    from oracle.oracle import Oracle
    # (the oracle's sparse-pattern products give bit-identical numbers 7 x faster for the reference's own sparse problems: cnot3)
    sp = bool(getattr(p, "use_sparse", False))
    r = Oracle(p, use_sparse=sp).traceobjgrad(pcof)
    objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, p, wa, False, True)
    gn = np.linalg.norm(r["totalgrad"])
    assert abs(prim - r["primaryobjf"]) <= tol * max(abs(r["primaryobjf"]), 1e-4)
    assert abs(sec - r["secondaryobjf"]) <= tol * abs(r["secondaryobjf"]), (sec, r["secondaryobjf"])
    assert np.linalg.norm(tg - r["totalgrad"]) <= tol * gn
    assert np.linalg.norm(ig - r["infidelgrad"]) <= tol * gn
    if p.objFuncType != 1:
        assert np.linalg.norm(lg - r["leakgrad"]) <= tol * gn
    if family is not None:
        assert wa.last_timing()["kernel_family"] == family, wa.last_timing()
    # forward only
    o2 = jq.traceobjgrad(pcof, p, wa, False, False)
    assert abs(o2[0] - r["objfv"]) <= tol * abs(r["objfv"])
    for nq in ensembles:
        nodes = 0.05 * rng.standard_normal(nq)
        weights = rng.random(nq)
        shift = rng.standard_normal(p.Ntot) * 0.05
        shift[0] = 0.0
        ref = Oracle(p, use_sparse=sp).eval_f_g_grad(pcof, nodes, weights, shift)
        jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=shift)
        gref = np.linalg.norm(ref["last_infidelity_grad"])
        assert abs(p.last_infidelity - ref["last_infidelity"]) <= tol * abs(ref["last_infidelity"])
        assert abs(p.last_leak - ref["last_leak"]) <= tol * abs(ref["last_leak"])
        assert np.linalg.norm(p.last_infidelity_grad - ref["last_infidelity_grad"]) <= tol * gref
        if p.objFuncType != 1:
            assert np.linalg.norm(p.last_leak_grad - ref["last_leak_grad"]) <= tol * gref
    return r


@pytest.mark.parametrize("oft", [1, 3])
@pytest.mark.parametrize("complex_states", [False, True])
def test_swap02_row_lane(jq, oft, complex_states):
    p, pcof = forbidden_problem("swap02", 2, 21, complex_states, oft, nsteps=1501)      # (the oracle loops over 600 samples)
    wa = make_wa(jq, p, pcof.size)
    compare(jq, p, pcof, wa, family=3, ensembles=(1, 7, 600), rng=np.random.default_rng(1))
    wa.close()


@pytest.mark.parametrize("oft,env,family", [(1, {}, 3), (2, {}, 3), (1, {"JQ_LANE": "0"}, 6), (3, {"JQ_LANE": "0", "JQ_EMBED": "0"}, 0)])
def test_cnot2(jq, oft, env, family):
    """row-lane kernels (NPJ = 12); JQ_LANE=0: the embedded 4 x 4 x 1 twin on the quad-layout kernels with the low-rank terms;
    JQ_LANE=0 JQ_EMBED=0: the dense NT = 1 slab kernels (run-time test of a.wrank)"""
    p, pcof = forbidden_problem("cnot2", 3, 22, True, oft, nsteps=1201)
    wa = make_wa(jq, p, pcof.size, env)
    compare(jq, p, pcof, wa, family=family, ensembles=(5, 70), rng=np.random.default_rng(2))
    wa.close()


@pytest.mark.parametrize("oft,nforb,env,family,band", [
    (1, 2, {"JQ_CQ_W": "0"}, 6, 7), (3, 4, {}, 6, 7),                     # quad layout, WLRT instantiation (complex rank 4: more than four slots)
    (1, 2, {}, 8, 7), (2, 2, {}, 8, 7),                                   # round 5: complex rank <= 2 on the cooperative-quad kernels (split backward sweep)
    (1, 2, {"JQ_T4": "0"}, 1, 9), (2, 2, {"JQ_T4": "0", "JQ_COOP_MAX": "0"}, 1, 9),      # JQ_BW_OD: cooperative kernels, whatever JQ_COOP_MAX says
    (1, 3, {"JQ_T4": "0", "JQ_OD": "0"}, 1, 1),                           # block-tridiagonal band tiles
    (3, 2, {"JQ_FORCE_DENSE": "1", "JQ_EMBED": "0"}, 1, 5),               # dense 96 x 96 tiles, small batches: cooperative kernels with HBM operands (round 6)
    (3, 2, {"JQ_FORCE_DENSE": "1", "JQ_EMBED": "0", "JQ_COOP_MAX": "0"}, 0, 5),      # ... large batches (here: forced): slab <6, 5> with the low-rank terms
])
def test_cnot3_short(jq, oft, nforb, env, family, band):
    p, pcof = forbidden_problem("cnot3", nforb, 23, True, oft, nsteps=300)
    wa = make_wa(jq, p, pcof.size, env)
    compare(jq, p, pcof, wa, family=family, ensembles=(3, 9), rng=np.random.default_rng(3))
    assert wa.last_timing()["kernel_band"] == band
    wa.close()


@pytest.mark.parametrize("cfg", [
    # Ntot, N, Nc, Nfreq, nsteps, m, objFuncType, banded, nforb
    (3, 3, 1, 2, 17, 2, 1, False, 1), (8, 3, 3, 2, 9, 2, 3, False, 8), (16, 16, 1, 1, 9, 1, 1, False, 5),
    (17, 5, 2, 2, 21, 4, 2, False, 2), (33, 7, 1, 3, 11, 5, 3, True, 3), (50, 8, 2, 1, 7, 7, 1, True, 16), (64, 4, 3, 2, 6, 3, 3, False, 1),
    (81, 9, 2, 1, 5, 1, 2, False, 4), (40, 3, 2, 2, 7, 3, 3, "od", 2), (48, 4, 4, 1, 6, 3, 1, "t4", 3), (36, 5, 3, 2, 7, 4, 2, "t4", 2),
    (112, 4, 3, 1, 5, 6, 1, "t4", 2), (130, 4, 2, 1, 5, 3, 3, True, 3), (200, 6, 1, 2, 4, 2, 1, False, 2), (40, 20, 2, 1, 6, 2, 1, False, 2),
], ids=lambda c: "Ntot%d_N%d_%s_r%d" % (c[0], c[1], c[7] if isinstance(c[7], str) else ("band" if c[7] else "dense"), c[8]))
def test_random_problems(jq, cfg):
    """sizes, structures and ranks the reference cases do not reach: every tile count, ragged slabs, N > 16, Ntot > 96 (cooperative
    kernels with the operators read from HBM), rank up to JQ_MAX_WRANK, more slabs than one round of cooperative workgroups"""
    Ntot, N, Nc, Nfreq, nsteps, m, oft, banded, nforb = cfg
    rng = np.random.default_rng(77 + Ntot)
    p, pcof = random_problem(jq, rng, Ntot, N, Nc, Nfreq, nsteps, m, oft, banded)
    set_forbidden(p, rng, nforb, complex_states=(Ntot % 2 == 0))
    wa = make_wa(jq, p, pcof.size)
    sps = max(16 // N, 1)
    ens = (sps + 1, 3 * sps + 2) + ((300 * sps + 1,) if Ntot == 33 else ())
    compare(jq, p, pcof, wa, ensembles=ens, rng=rng)
    if Ntot == 33:
        assert wa.last_timing()["kernel_family"] == 1      # 301 slabs on the cooperative kernels: the slab kernels have no low-rank terms
    wa.close()


def test_negative_weights_and_rank_detection(jq):
    """forb_weights may be negative (an indefinite Hermitian W); linearly dependent forbidden states lower the rank"""
    p, pcof = forbidden_problem("cnot2", 2, 24, True, 1)
    rng = np.random.default_rng(5)
    f = rng.standard_normal((p.Ntot, 3)) + 1j * rng.standard_normal((p.Ntot, 3))
    f[:, 2] = f[:, 0] + 0.5j * f[:, 1]                                 # rank 2 from three states
    W = 0.8 * np.outer(f[:, 0], f[:, 0].conj()) - 0.3 * np.outer(f[:, 1], f[:, 1].conj()) + 0.1 * np.outer(f[:, 2], f[:, 2].conj())
    p.wmat_real, p.wmat_imag = np.asfortranarray(W.real.copy()), np.asfortranarray(W.imag.copy())
    wa = make_wa(jq, p, pcof.size)
    compare(jq, p, pcof, wa, family=3)
    wa.close()


def test_full_diagonal_matrix_takes_the_diagonal_path_bit_for_bit(jq):
    p, info, pcof, _ = case_inputs("cnot2")
    wa = make_wa(jq, p, pcof.size)
    a = jq.traceobjgrad(pcof, p, wa, False, True)
    w0 = p.wmat_real.copy()
    p.wmat_real = np.diag(w0)
    p.wmat_imag = np.zeros_like(p.wmat_real)
    b = jq.traceobjgrad(pcof, p, wa, False, True)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    # full weights and back: the Diagonal results return exactly
    set_forbidden(p, np.random.default_rng(6), 2)
    c = jq.traceobjgrad(pcof, p, wa, False, True)
    assert c[0] != a[0]
    p.wmat_real = w0
    p.wmat_imag = np.zeros(p.Ntot)
    d = jq.traceobjgrad(pcof, p, wa, False, True)
    assert a[0] == d[0] and np.array_equal(a[1], d[1])
    wa.close()


def test_refusals(jq):
    from juqbox_jl_amd import _lib
    p, pcof = forbidden_problem("swap02", 2, 25, True, 1)
    good_r, good_i = p.wmat_real.copy(), p.wmat_imag.copy()
    wa = make_wa(jq, p, pcof.size)
    jq.traceobjgrad(pcof, p, wa, False, True)
    # not Hermitian: asymmetric real part / symmetric imaginary part
    for wr, wi in ((good_r + np.triu(np.ones_like(good_r), 1) * 1e-3, good_i), (good_r, good_i + 1e-3 * np.eye(p.Ntot))):
        p.wmat_real, p.wmat_imag = wr, wi
        with pytest.raises(_lib.JuqboxHipError) as e:
            jq.traceobjgrad(pcof, p, wa, False, True)
        assert e.value.code == _lib.JQ_EUNSUPPORTED and "Hermitian" in str(e.value)
    # (round 6: the Jacobi solver and ranks above 16 are no longer refused -- tests/test_gpu_round6.py; here: the switch on a live handle)
    p.wmat_real, p.wmat_imag = good_r, good_i
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER, max_iter=50, tol=1e-12, nrhs=p.N)
    compare(jq, p, pcof, wa)
    assert wa.plan_info()["linear_solver"] == "jacobi" and wa.plan_info()["embedded_twin_Ntot"] == 0
    wa.close()
    rng = np.random.default_rng(8)
    # the implicit-midpoint type never sees wmat_real: its weights are params.wmat (Diagonal), results unchanged by forb_states
    p3, info3, pcof3, _ = case_inputs("swap02")
    p3.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12)
    wm = jq.Working_Arrays_M_HIP(p3, pcof3.size)
    a = jq.traceobjgrad(pcof3, p3, wm, False, True)
    set_forbidden(p3, rng, 2)
    b = jq.traceobjgrad(pcof3, p3, wm, False, True)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    h = wm.handle
    L = _lib.load()
    wr = np.ascontiguousarray(p3.wmat_real.ravel(order="F"))
    assert L.jq_update_wmat(h, wr.ctypes.data_as(_lib.c_dp), None) == _lib.JQ_EUNSUPPORTED
    wm.close()


@pytest.mark.parametrize("complex_states", [True, False])
def test_multi_device_handle_and_replanning_carry_the_weights(jq, complex_states):
    """(real forbidden states, round 5: the sub-handles' shards run on the cooperative-quad kernels with the low-rank terms)"""
    with jq.options(multi_same_device=1):
        p, pcof = forbidden_problem("cnot3", 2, 26, complex_states, 3, nsteps=200)
        wa1 = make_wa(jq, p, pcof.size)
        wa3 = make_wa(jq, p, pcof.size, devices=3)
        rng = np.random.default_rng(9)
        nodes, weights = 1e-3 * rng.standard_normal(11), rng.random(11)
        shift = np.concatenate([[0.0], rng.standard_normal(p.Ntot - 1)])      # (the reference's 0.01 * 10^(j-2) overflows at Ntot = 96)
        jq.eval_f_g_grad(pcof, p, wa1, nodes, weights, True, shift=shift)
        assert wa1.last_timing()["kernel_family"] == 8      # (round 5: four slots on the cooperative-quad kernels, real or complex)
        ref = (p.last_infidelity, p.last_leak, p.last_infidelity_grad.copy(), p.last_leak_grad.copy())
        jq.eval_f_g_grad(pcof, p, wa3, nodes, weights, True, shift=shift)
        assert abs(p.last_infidelity - ref[0]) <= 1e-13 * abs(ref[0]) and abs(p.last_leak - ref[1]) <= 1e-13 * abs(ref[1])
        assert np.linalg.norm(p.last_infidelity_grad - ref[2]) <= 1e-12 * np.linalg.norm(ref[2])
        assert np.linalg.norm(p.last_leak_grad - ref[3]) <= 1e-12 * np.linalg.norm(ref[2])
        wa3.close()
        # a drift outside the 4 x 4 x n structure re-plans the handle (dense tiles): the full weights come along
        p.Hconst = p.Hconst.copy()
        p.Hconst[0, 37] = p.Hconst[37, 0] = 1e-3
        compare(jq, p, pcof, wa1)
        assert wa1.plan_info()["replanned"] is True
        wa1.close()


@pytest.mark.parametrize("case", ["swap02", "cnot2", "cnot3"])
def test_csc_entry_is_bit_identical_to_the_dense_entry(jq, case):
    """CSC operator storage (colptr / rowval / nzval, 1-based Int64) for use_sparse = true problems"""
    p, info, pcof, golden = case_inputs(case)
    if case == "cnot3":
        p.T, p.nsteps = p.T * 500 / p.nsteps, 500
    wd = make_wa(jq, p, pcof.size, csc=False)
    ws = make_wa(jq, p, pcof.size, csc=True)
    assert ws.plan_info() == wd.plan_info()
    a = jq.traceobjgrad(pcof, p, wd, False, True)
    b = jq.traceobjgrad(pcof, p, ws, False, True)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[5], b[5])
    # drift update through the sparse entry
    p.Hconst = p.Hconst.copy()
    for j in range(1, p.Ntot):
        p.Hconst[j, j] += 1e-3 * j
    a = jq.traceobjgrad(pcof, p, wd, False, True)
    b = jq.traceobjgrad(pcof, p, ws, False, True)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    wd.close()
    ws.close()


def test_csc_errors(jq):
    import ctypes
    from juqbox_jl_amd import _lib
    from juqbox_jl_amd.evalobjgrad import _Csc
    p, info, pcof, _ = case_inputs("swap02")
    wa = make_wa(jq, p, pcof.size, csc=True)
    L = _lib.load()
    k = _Csc(p.Hconst)
    k.rowval[0] = p.Ntot + 1
    assert L.jq_update_hconst_csc(wa.handle, ctypes.byref(k.desc)) == _lib.JQ_EINVAL
    k = _Csc(p.Hconst)
    k.colptr[0] = 0
    assert L.jq_update_hconst_csc(wa.handle, ctypes.byref(k.desc)) == _lib.JQ_EINVAL
    k = _Csc(np.zeros((p.Ntot + 1, p.Ntot + 1)))
    assert L.jq_update_hconst_csc(wa.handle, ctypes.byref(k.desc)) == _lib.JQ_EINVAL
    assert L.jq_update_hconst_csc(wa.handle, None) == _lib.JQ_EINVAL
    wa.close()
