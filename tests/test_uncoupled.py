"""Uncoupled controls Hunc_ops = the LAB-frame evaluation of a pulse (KS!, src/evalobjgrad.jl:2373-2387; SURVEY.md section 8
rows a6 / f4).  The reference holds no golden for this branch (parity unpinned).  Pinned here:
 CPU  the oracle's forward restatement against PHYSICS: the same optimised pulse (examples/drives/cnot2-pcof-opt-t50.jld2)
      evaluated in the rotating frame (coupled controls, rotated target) and in the lab frame (uncoupled controls, unrotated
      target, ft = 2 (p cos(2 pi f t) - q sin(2 pi f t))) must give the same gate infidelity up to the rotating-wave error;
 CPU  the oracle's adjoint gradient against central finite differences of its own objective.  The reference's adjoint for this
      branch is not a usable reference: adjoint_grad_calc! (:2620-2656) differentiates control functions of an older numbering
      (func = 2 Ncoupled - 1 + q, no rotation factor) -- not what KS! applies -- and throws for objFuncType != 1 (gradSize,
      :801); the oracle and the device compute the gradient of the objective that the forward sweep evaluates;
 GPU  the HIP path against the oracle (objective, per-step states, gradient) at 1e-10."""
import os

import numpy as np
import pytest
from conftest import GOLDEN_DIR, case_inputs

DRIVE = os.path.join(GOLDEN_DIR, "jld2", "cnot2-pcof-opt-t50.jld2")


def rotating_frame_twin(jq, lab, rot_freq):
    """the rotating-frame problem of examples/cnot2-setup.jl for the same levels / gate time as the lab-frame one"""
    from juqbox_jl_amd import setup_utils as su
    a, b = lab.Hunc_ops
    amat, bmat = np.triu(a), np.triu(b)
    Ntot = lab.Ntot
    k = np.arange(Ntot)
    N1, N2 = np.diag((k % 3).astype(float)), np.diag((k // 3).astype(float))
    H0 = lab.Hconst - 2 * np.pi * (rot_freq[0] * N1 + rot_freq[1] * N2)
    om1, om2 = su.setup_rotmatrices(lab.Ne, lab.Ng, rot_freq)
    utarget = lab.Utarget_r + 1j * lab.Utarget_i
    vtarget = (np.exp(1j * om1 * lab.T) * np.exp(1j * om2 * lab.T))[:, None] * utarget
    return jq.objparams(lab.Ne, lab.Ng, lab.T, lab.nsteps, Uinit=lab.Uinit, Utarget=vtarget, Cfreq=lab.Cfreq, Rfreq=rot_freq,
                        Hconst=H0, Hsym_ops=[amat + amat.T, bmat + bmat.T], Hanti_ops=[amat - amat.T, bmat - bmat.T])


def test_oracle_lab_frame_agrees_with_the_rotating_frame_up_to_the_rotating_wave_error(jq):
    from oracle.oracle import Oracle
    lab, info = jq.cases.cnot2_lab(Pmin=80, pcof_file=DRIVE)
    pcof = info["pcof0"]
    assert lab.Nunc == 2 and lab.Ncoupled == 0 and lab.isSymm == [True, True]
    r_lab = Oracle(lab).traceobjgrad(pcof, evaladjoint=False)
    rot = rotating_frame_twin(jq, lab, info["rot_freq"])
    r_rot = Oracle(rot).traceobjgrad(pcof, evaladjoint=False)
    # an optimised CNOT pulse: small infidelity in its own (rotating) frame ...
    assert r_rot["traceInfidelity"] < 2e-2
    # ... and the lab-frame evaluation sees the same gate up to the counter-rotating terms (amplitude / frequency ~ 5e-3)
    assert abs(r_lab["traceInfidelity"] - r_rot["traceInfidelity"]) < 2e-2
    # the factor 2 and the sin/cos signs of ft matter: a wrong rotation frequency destroys the gate
    lab.Rfreq = lab.Rfreq * 1.01
    assert Oracle(lab).traceobjgrad(pcof, evaladjoint=False)["traceInfidelity"] > 0.3


def lab_problem(jq, Pmin, anti, oft=1):
    lab, info = jq.cases.cnot2_lab(Pmin=Pmin)
    if anti:      # the second control antisymmetric: its term goes to S (the Neumann solves see it)
        b = lab.Hunc_ops[1]
        lab.Hunc_ops[1] = np.triu(b) - np.triu(b).T
        lab.isSymm[1] = False
        lab.linear_solver.max_iter = 4
    lab.objFuncType = oft
    return lab, info["pcof0"]


@pytest.mark.parametrize("anti", [False, True])
def test_oracle_gradient_for_uncoupled_controls_is_the_derivative_of_its_objective(jq, anti):
    """central finite differences of the discrete objective in six coefficient directions (cos and sin blocks of both
    controls) -- the adjoint gradient is exact for the discrete scheme, so the agreement is limited by the differences"""
    from oracle.oracle import Oracle
    lab, pcof = lab_problem(jq, 5, anti)
    rng = np.random.default_rng(5)
    pcof = pcof + 1e-3 * rng.standard_normal(pcof.size)
    o = Oracle(lab)
    g = o.traceobjgrad(pcof, evaladjoint=True)["totalgrad"]
    assert np.linalg.norm(g) > 1e-6
    eps = 1e-6
    for i in rng.choice(pcof.size, 6, replace=False):
        d = np.zeros(pcof.size)
        d[i] = eps
        fd = (o.traceobjgrad(pcof + d, evaladjoint=False)["objfv"] - o.traceobjgrad(pcof - d, evaladjoint=False)["objfv"]) / (2 * eps)
        assert abs(fd - g[i]) <= 1e-6 * np.linalg.norm(g, np.inf) + 1e-9, (i, fd, g[i])


def test_operators_that_are_neither_symmetric_nor_antisymmetric_are_the_references_argument_error(jq):
    lab, info = jq.cases.cnot2_lab(Pmin=5)
    bad = lab.Hunc_ops[0].copy()
    bad[0, 1] += 0.5
    with pytest.raises(ValueError, match="not symmetric or anti-symmetric"):
        jq.objparams(lab.Ne, lab.Ng, lab.T, lab.nsteps, Uinit=lab.Uinit, Utarget=lab.Utarget_r + 0j, Cfreq=lab.Cfreq,
                     Rfreq=lab.Rfreq, Hconst=lab.Hconst, Hunc_ops=[bad])


@pytest.mark.gpu
@pytest.mark.parametrize("anti", [False, True])
def test_gpu_lab_frame_evaluation_matches_the_oracle(jq, anti):
    """objective + per-step states through the C ABI; `anti`: the second control antisymmetric (its term goes to S: the
    Neumann solves are exercised), 36 000+ time steps = two chunks"""
    from juqbox_jl_amd import _lib
    from oracle.oracle import Oracle
    lab, info = jq.cases.cnot2_lab(Pmin=40, pcof_file=DRIVE)
    pcof = info["pcof0"]
    if anti:
        b = lab.Hunc_ops[1]
        lab.Hunc_ops[1] = np.triu(b) - np.triu(b).T
        lab.isSymm[1] = False
        lab.linear_solver.max_iter = 4
    r = Oracle(lab).traceobjgrad(pcof, evaladjoint=False, history=True)
    wa = jq.Working_Arrays_HIP(lab, pcof.size)
    objfv, prim, sec = jq.traceobjgrad(pcof, lab, wa, False, False)
    assert abs(objfv - r["objfv"]) <= 1e-10 * abs(r["objfv"])
    assert abs(sec - r["secondaryobjf"]) <= 1e-10 * abs(r["secondaryobjf"]) + 1e-18
    objv, hist, fid = jq.traceobjgrad(pcof, lab, wa, True, False)
    assert np.max(np.abs(hist - r["history"])) < 1e-10         # 36 000 steps of a GHz-frequency evolution
    # ensembles / sweeps are forward evaluations too
    sw = jq.traceobj_sweep(pcof, lab, wa, np.array([0.0, 1e-3]), np.arange(lab.Ntot) * 1.0)
    assert abs(sw[0, 0] - objfv) <= 1e-12
    # gradient (Stormer-Verlet): against the oracle's, which finite differences pin above
    g = Oracle(lab).traceobjgrad(pcof, evaladjoint=True)
    objfv2, tg, prim2, sec2, tinf, ig, lg = jq.traceobjgrad(pcof, lab, wa, False, True)
    assert abs(objfv2 - g["objfv"]) <= 1e-10 * abs(g["objfv"])
    assert np.linalg.norm(tg - g["totalgrad"]) <= 1e-9 * np.linalg.norm(g["totalgrad"])
    # an ensemble with the gradient (two nodes)
    nodes, weights, shift = np.array([-1e-3, 2e-3]), np.array([0.4, 0.6]), np.arange(lab.Ntot) * 1.0
    ref = Oracle(lab).eval_f_g_grad(pcof, nodes, weights, shift)
    jq.eval_f_g_grad(pcof, lab, wa, nodes, weights, True, shift=shift)
    assert abs(lab.last_infidelity - ref["last_infidelity"]) <= 1e-10 * abs(ref["last_infidelity"])
    assert np.linalg.norm(lab.last_infidelity_grad - ref["last_infidelity_grad"]) <= 1e-9 * np.linalg.norm(ref["last_infidelity_grad"])
    wa.close()


@pytest.mark.gpu
def test_gpu_gradient_for_uncoupled_controls_with_the_leak_constraint_split(jq):
    """objFuncType = 3 (forced and unforced backward sweep): infidelity and leak gradients against the oracle (the reference
    throws here, gradSize :801)"""
    from oracle.oracle import Oracle
    lab, pcof = lab_problem(jq, 10, True, oft=3)
    g = Oracle(lab).traceobjgrad(pcof, evaladjoint=True)
    wa = jq.Working_Arrays_HIP(lab, pcof.size)
    objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, lab, wa, False, True)
    gn = np.linalg.norm(g["totalgrad"])
    assert np.linalg.norm(tg - g["totalgrad"]) <= 1e-9 * gn
    assert np.linalg.norm(ig - g["infidelgrad"]) <= 1e-9 * gn
    assert np.linalg.norm(lg - g["leakgrad"]) <= 1e-9 * gn
    wa.close()


@pytest.mark.gpu
def test_gpu_implicit_midpoint_gradient_for_uncoupled_controls_is_refused(jq):
    from juqbox_jl_amd import _lib
    lab, pcof = lab_problem(jq, 5, False)
    lab.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=lab.N)
    wa = jq.Working_Arrays_M_HIP(lab, pcof.size)
    with pytest.raises(_lib.JuqboxHipError) as e:
        jq.traceobjgrad(pcof, lab, wa, False, True)
    assert e.value.code == _lib.JQ_EUNSUPPORTED and "implicit-midpoint" in str(e.value)
    wa.close()
