"""Full / complex leakage weights (`use_custom_forbidden`, src/evalobjgrad.jl:214-232; objective :700, :716-718 with
penalf2a / penalf2aTrap full :2183-2223 and penalf2imag :2226-2228; adjoint forcing :862, :882-888).

PARITY-UNPINNED IN THE REFERENCE: no test, example or golden of the reference uses this branch.  The oracle restates the cited
lines; what pins it here is (i) equality with the pinned Diagonal path when the full matrix is a diagonal one, (ii) the leak
integral against an independent numpy evaluation of psi^H W psi on the oracle's own state history, (iii) central finite
differences of its objective against its adjoint gradient.  The GPU tests (tests/test_gpu_dense_wmat.py) compare the device with
this oracle at 1e-10."""
import numpy as np
import pytest

from conftest import case_inputs
from oracle.oracle import Oracle


def forbidden_problem(case, nforb, seed, complex_states=True, objFuncType=1, nsteps=None):
    """The named set-up with `nforb` random forbidden states (weights 0.5 .. 1.5) in place of the Diagonal guard-level weights."""
    params, info, pcof, _ = case_inputs(case)
    rng = np.random.default_rng(seed)
    fs = rng.standard_normal((params.Ntot, nforb)) + (1j * rng.standard_normal((params.Ntot, nforb)) if complex_states else 0)
    fs = fs / np.linalg.norm(fs, axis=0)
    fw = 0.5 + rng.random(nforb)
    W = sum(fw[k] * np.outer(fs[:, k], np.conj(fs[:, k])) for k in range(nforb))
    params.forb_states, params.forb_weights = fs, fw
    params.wmat_real = np.asfortranarray(W.real.copy())
    params.wmat_imag = np.asfortranarray(W.imag.copy())
    params.objFuncType = objFuncType
    if nsteps is not None:
        params.T = params.T * nsteps / params.nsteps
        params.nsteps = nsteps
    return params, pcof


def test_constructor_builds_the_weight_matrices():
    import juqbox_jl_amd as jq
    base, _ = jq.cases.swap02()
    rng = np.random.default_rng(3)
    fs = rng.standard_normal((base.Ntot, 2)) + 1j * rng.standard_normal((base.Ntot, 2))
    fw = np.array([0.7, 1.3])
    p = jq.objparams(base.Ne, base.Ng, base.T, base.nsteps, Uinit=base.Uinit, Utarget=base.Utarget_r + 1j * base.Utarget_i,
                     Cfreq=base.Cfreq, Rfreq=base.Rfreq, Hconst=base.Hconst, Hsym_ops=base.Hsym_ops, Hanti_ops=base.Hanti_ops,
                     use_custom_forbidden=True, forb_states=fs, forb_weights=fw)
    # src/evalobjgrad.jl:222-231: wmat[i,j] += w * conj(f_j) * f_i
    W = np.zeros((base.Ntot, base.Ntot), dtype=complex)
    for k in range(2):
        for j in range(base.Ntot):
            for i in range(base.Ntot):
                W[i, j] += fw[k] * np.conj(fs[j, k]) * fs[i, k]
    assert np.allclose(p.wmat_real, W.real, atol=1e-15) and np.allclose(p.wmat_imag, W.imag, atol=1e-15)
    assert np.allclose(p.wmat_real, p.wmat_real.T) and np.allclose(p.wmat_imag, -p.wmat_imag.T)
    with pytest.raises(ValueError):
        jq.objparams(base.Ne, base.Ng, base.T, base.nsteps, Uinit=base.Uinit, Utarget=base.Utarget_r + 1j * base.Utarget_i,
                     Cfreq=base.Cfreq, Rfreq=base.Rfreq, Hconst=base.Hconst, Hsym_ops=base.Hsym_ops, Hanti_ops=base.Hanti_ops,
                     use_custom_forbidden=True, forb_states=fs[:-1], forb_weights=fw)


@pytest.mark.parametrize("case", ["swap02", "cnot2"])
def test_full_diagonal_matrix_equals_the_diagonal_path(case):
    params, info, pcof, _ = case_inputs(case)
    ref = Oracle(params).traceobjgrad(pcof)
    params.wmat_real = np.diag(params.wmat_real)
    params.wmat_imag = np.zeros_like(params.wmat_real)
    r = Oracle(params).traceobjgrad(pcof)
    assert abs(r["objfv"] - ref["objfv"]) <= 1e-13 * abs(ref["objfv"])
    assert np.linalg.norm(r["totalgrad"] - ref["totalgrad"]) <= 1e-12 * np.linalg.norm(ref["totalgrad"])


@pytest.mark.parametrize("complex_states", [False, True])
def test_leak_integral_against_the_state_history(complex_states):
    params, pcof = forbidden_problem("swap02", 2, 11, complex_states)
    r = Oracle(params).traceobjgrad(pcof, evaladjoint=False, history=True)
    # independent evaluation: vr = Re psi, vi = -Im psi at the integer time points; the half-step values vi05 are not in the
    # history, so this checks the real-weight trapezoidal part exactly and the rest to O(dt)
    h = r["history"]
    Wr = params.wmat_real
    dt = params.T / params.nsteps
    quad = np.einsum("ijn,ik,kjn->n", h.real, Wr, h.real) + np.einsum("ijn,ik,kjn->n", h.imag, Wr, h.imag)
    W = params.wmat_real + 1j * params.wmat_imag
    full = np.einsum("ijn,ik,kjn->n", np.conj(h), W, h).real          # psi^H W psi
    trap = dt / params.T * (0.5 * full[0] + full[1:-1].sum() + 0.5 * full[-1])
    assert abs(trap - r["secondaryobjf"]) < 5e-3 * abs(trap) + 1e-12
    assert np.all(quad > -1e-14)


@pytest.mark.parametrize("case,nforb,complex_states,oft", [("swap02", 1, True, 1), ("swap02", 3, True, 3), ("swap02", 2, False, 2),
                                                            ("cnot2", 2, True, 1)])
def test_adjoint_gradient_against_finite_differences(case, nforb, complex_states, oft):
    params, pcof = forbidden_problem(case, nforb, 5, complex_states, oft, nsteps=600 if case == "cnot2" else None)
    o = Oracle(params)
    r = o.traceobjgrad(pcof)
    rng = np.random.default_rng(0)
    worst = 0.0
    for i in rng.choice(pcof.size, 6, replace=False):
        e = np.zeros_like(pcof)
        hstep = 1e-5 * max(1.0, abs(pcof[i]))
        e[i] = hstep
        fp = o.traceobjgrad(pcof + e, evaladjoint=False)
        fm = o.traceobjgrad(pcof - e, evaladjoint=False)
        for key, g in (("objfv", r["totalgrad"]), ("primaryobjf", r["infidelgrad"]), ("secondaryobjf", r["leakgrad"])):
            if oft == 1 and key != "objfv":
                continue
            fd = (fp[key] - fm[key]) / (2 * hstep)
            worst = max(worst, abs(fd - g[i]) / np.linalg.norm(g) * np.sqrt(pcof.size))
    # the adjoint is the exact gradient of the discrete objective: the error is the noise of the difference quotient (the pinned
    # Diagonal path reaches the same 1e-8 .. 1e-7 with this step)
    assert worst < 1e-6, worst
