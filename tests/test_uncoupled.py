"""Uncoupled controls Hunc_ops = the LAB-frame evaluation of a pulse (KS!, src/evalobjgrad.jl:2373-2387; SURVEY.md section 8
rows a6 / f4).  The reference holds no golden for this branch (parity unpinned) and its adjoint for it cannot run
(gradSize, :801, is not length(pcof)), so the branch is forward-only here and pinned twice:
 CPU  the oracle's restatement against PHYSICS: the same optimised pulse (examples/drives/cnot2-pcof-opt-t50.jld2) evaluated in
      the rotating frame (coupled controls, rotated target) and in the lab frame (uncoupled controls, unrotated target,
      ft = 2 (p cos(2 pi f t) - q sin(2 pi f t))) must give the same gate infidelity up to the rotating-wave error;
 GPU  the HIP path against the oracle (objective, per-step states) at 1e-10."""
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


def test_oracle_refuses_the_adjoint_for_uncoupled_controls_like_the_reference_would_throw(jq):
    from oracle.oracle import Oracle
    lab, info = jq.cases.cnot2_lab(Pmin=5)
    with pytest.raises(ValueError, match="DimensionMismatch"):
        Oracle(lab).traceobjgrad(info["pcof0"], evaladjoint=True)


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
    # gradients: refused, with the reason
    with pytest.raises(_lib.JuqboxHipError) as e:
        jq.traceobjgrad(pcof, lab, wa, False, True)
    assert e.value.code == _lib.JQ_EUNSUPPORTED and "gradSize" in str(e.value)
    wa.close()
