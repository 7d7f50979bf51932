"""CPU: pins the oracle (oracle/juqbox_oracle.c) against every Stormer-Verlet golden vector of the
reference's test-suite (test/runtests.jl:30-54 via test/evalGrad.jl), at the reference's tolerance."""
import numpy as np
import pytest
from conftest import case_inputs, load_golden, reference_pass

from oracle.oracle import Oracle

SV_CASES = ["rabi", "swap02", "flux", "cnot2", "cnot2-leakieq", "cnot2-jacobi", "cnot3"]


def eval_like_evalGrad(jq, params, pcof, r):
    """objv/grad exactly as test/evalGrad.jl:14-25 assembles them (Tikhonov terms included)."""
    su = jq.setup_utils
    tik = su.tikhonov_pen(pcof, params.tik0)
    tg = su.tikhonov_grad(pcof, params.tik0)
    if params.objFuncType == 1:
        return np.array([r["primaryobjf"] + r["secondaryobjf"] + tik]), r["infidelgrad"] + tg
    return (np.array([r["primaryobjf"] + tik, r["secondaryobjf"]]),
            np.concatenate([r["infidelgrad"] + tg, r["leakgrad"]]))


@pytest.mark.parametrize("case", SV_CASES)
def test_oracle_reproduces_reference_golden(jq, case):
    params, info, pcof, golden = case_inputs(case)
    r = Oracle(params).traceobjgrad(pcof)
    obj, grad = eval_like_evalGrad(jq, params, pcof, r)
    assert reference_pass(obj, golden["obj0"]), (obj, golden["obj0"])
    assert reference_pass(grad, golden["grad0"])


@pytest.mark.parametrize("case", SV_CASES)
def test_oracle_reproduces_reference_golden_implicit_midpoint(jq, case):
    """The second loop of test/runtests.jl:60-80: the same seven set-ups with params.Integrator_id = 2 and
    lsolver_object(solver=JACOBI_SOLVER_M, max_iter=100, tol=1e-12), against <case>-ref-imr.jld2
    (tests/golden/<case>-imr.json, tests/golden/make_golden_imr.py)."""
    params, info, pcof, _ = case_inputs(case)
    golden = load_golden(case + "-imr")
    r = Oracle(params).traceobjgrad_imr(pcof, golden["solver"]["max_iter"], golden["solver"]["tol"])
    obj, grad = eval_like_evalGrad(jq, params, pcof, r)
    assert reference_pass(obj, golden["obj0"]), (obj, golden["obj0"])
    assert reference_pass(grad, golden["grad0"])


def test_cnot3_decomposition_matches_survey(jq):
    """SURVEY.md section 8c records the decomposition of the cnot3 golden objective."""
    params, info, pcof, golden = case_inputs("cnot3")
    r = Oracle(params).traceobjgrad(pcof, evaladjoint=False)
    assert abs(r["primaryobjf"] - 0.9181500713381303) < 1e-13
    assert abs(r["secondaryobjf"] - 2.8775930168455916e-05) < 1e-16


def test_dense_and_sparse_products_agree(jq):
    params, info, pcof, golden = case_inputs("cnot2")
    a = Oracle(params, use_sparse=False).traceobjgrad(pcof)
    b = Oracle(params, use_sparse=True).traceobjgrad(pcof)
    assert abs(a["objfv"] - b["objfv"]) < 1e-13
    assert np.linalg.norm(a["totalgrad"] - b["totalgrad"]) < 1e-12 * np.linalg.norm(a["totalgrad"])


def test_time_reversibility_of_the_state(jq):
    """The adjoint sweep re-integrates the state backwards (src/evalobjgrad.jl:879); after it the
    state must be back at Uinit up to round-off and the Neumann truncation."""
    params, info, pcof, golden = case_inputs("swap02")
    r = Oracle(params).traceobjgrad(pcof, final_state=True)
    vr_back, vi_back = r["final_state"][:, :, 2], r["final_state"][:, :, 3]
    assert np.max(np.abs(vr_back - params.Uinit)) < 1e-9
    assert np.max(np.abs(vi_back)) < 1e-9


def test_input_validation_errors(jq):
    params, info, pcof, golden = case_inputs("swap02")
    o = Oracle(params)
    with pytest.raises(ValueError):      # src/evalobjgrad.jl:604-606 (Psize % Nsig != 0 || Psize < 3*Nsig)
        o.traceobjgrad(pcof[:3])
    with pytest.raises(ValueError):      # bcparams DimensionMismatch (src/bsplines.jl:178-181)
        o.traceobjgrad(np.concatenate([pcof, pcof[:2]]))


def test_ensemble_is_weighted_sum_of_single_evaluations(jq):
    """eval_f_g_grad! (src/ipopt_interface.jl:24-70) vs explicit loop with a mutated Hconst."""
    params, info, pcof, golden = case_inputs("swap02")
    nodes = np.array([-0.03, 0.0, 0.05])
    weights = np.array([0.25, 0.5, 0.25])
    shift = params.shift_weights_reference()
    ens = Oracle(params).eval_f_g_grad(pcof, nodes, weights, shift)
    inf = leak = 0.0
    grad = np.zeros(pcof.size)
    for ep, w in zip(nodes, weights):
        H = params.Hconst.copy()
        for j in range(1, params.Ntot):
            params.Hconst[j, j] += ep * shift[j]
        r = Oracle(params).traceobjgrad(pcof)
        params.Hconst[:] = H
        inf += w * r["primaryobjf"]
        leak += w * r["secondaryobjf"]
        grad += w * r["infidelgrad"]
    assert abs(ens["last_infidelity"] - inf) < 1e-14
    assert abs(ens["last_leak"] - leak) < 1e-15
    assert np.linalg.norm(ens["last_infidelity_grad"] - grad) < 1e-12 * np.linalg.norm(grad)


# --- known-answer test of the basic scheme: test/test-stormer-verlet.jl -------------------------
def _sv_step_alloc(K, S, t, u, v, h, uforce, vforce):
    """Juqbox.step with function forcing and Gaussian elimination (src/StormerVerlet.jl:37-91)."""
    In = np.eye(2)
    uforce0, vforce05, uforce1 = uforce(t), vforce(t + 0.5 * h), uforce(t + h)
    K0, S0, K05, S05, K1, S1 = K(t), S(t), K(t + 0.5 * h), S(t + 0.5 * h), K(t + h), S(t + h)
    rhs = K05 @ u + S05 @ v + vforce05
    l1 = np.linalg.solve(In - 0.5 * h * S05, rhs)
    v05 = v + 0.5 * h * l1
    kappa1 = S0 @ u - K0 @ v05 + uforce0
    rhs = S1 @ (u + 0.5 * h * kappa1) - K1 @ v05 + uforce1
    kappa2 = np.linalg.solve(In - 0.5 * h * S1, rhs)
    u = u + 0.5 * h * (kappa1 + kappa2)
    l2 = K05 @ u + S05 @ v05 + vforce05
    v = v + 0.5 * h * (l1 + l2)
    return t + h, u, v


def _timesteptest(cfl, testcase):
    """test/test-stormer-verlet.jl:7-135"""
    if testcase in (1, 2):
        K0m, S0m = np.array([[0.0, 1.0], [1.0, 0.0]]), np.zeros((2, 2))
    else:
        K0m, S0m = np.zeros((2, 2)), np.array([[0.0, 1.0], [-1.0, 0.0]])
    T = 5 * np.pi
    omega = 2 * np.pi
    maxeig = np.max(np.abs(np.linalg.eigvals(K0m + S0m)))
    dt = cfl / maxeig
    nsteps = int(np.ceil(T / dt))
    dt = T / nsteps
    z = lambda t: np.zeros(2)
    phi1 = lambda t: 0.25 * (t - np.sin(omega * t) / omega)
    phidot = lambda t: 0.5 * (np.sin(0.5 * omega * t)) ** 2
    quad = lambda t: 4 / T ** 2 * t * (T - t)
    if testcase == 1:
        timefunc, uforce, vforce = (lambda t: 0.25 * (1.0 - np.cos(omega * t))), z, z
    elif testcase == 0:
        timefunc, uforce, vforce = (lambda t: 0.25 * (1 - np.sin(omega * t))), z, z
    elif testcase == 2:
        timefunc = quad
        uforce = lambda t: np.array([(quad(t) - phidot(t)) * np.sin(phi1(t)), 0.0])
        vforce = lambda t: np.array([0.0, -(quad(t) - phidot(t)) * np.cos(phi1(t))])
    else:
        timefunc = quad
        uforce = lambda t: np.array([-phidot(t) * np.sin(phi1(t)), quad(t) * np.cos(phi1(t))])
        vforce = lambda t: np.array([-quad(t) * np.sin(phi1(t)), phidot(t) * np.cos(phi1(t))])
    K = lambda t: timefunc(t) * K0m
    S = lambda t: timefunc(t) * S0m
    u, v, t = np.array([1.0, 0.0]), np.array([0.0, 0.0]), 0.0
    for _ in range(nsteps):
        t, u, v = _sv_step_alloc(K, S, t, u, v, dt, uforce, vforce)
    if testcase in (1, 2, 3):
        phi = 0.25 * (t - 1.0 / omega * np.sin(omega * t))
        cg, ce = np.cos(phi), -1j * np.sin(phi)
    else:
        phi = 0.25 * (t + 1 / omega * (np.cos(omega * t) - 1.0))
        cg, ce = np.cos(phi), -np.sin(phi) + 0j
    cg_err = np.sqrt((u[0] - np.real(cg)) ** 2 + (v[0] + np.imag(cg)) ** 2)
    ce_err = np.sqrt((u[1] - np.real(ce)) ** 2 + (v[1] + np.imag(ce)) ** 2)
    return cg_err, ce_err


def test_ensemble_loop_perturbs_levels_whose_drift_entry_is_zero(jq):
    """round 5: `params.Hconst[j,j] += ...` on a SparseMatrixCSC (src/ipopt_interface.jl:41-44) is setindex! -- an entry that is not
    stored yet is inserted, and accumulate_matrix! (src/evalobjgrad.jl:2428-2440) carries it into K.  The oracle's sparse mode fixes
    its patterns at creation; its ensemble loop used to drop the perturbation of every level with Hconst[j,j] == 0 (cnot3 in the
    rotating frame has four): 8e-5 relative in the infidelity, found when a GPU ensemble test disagreed with it.  Sparse mode, dense
    mode and the explicit loop over freshly built oracles must agree."""
    params, info = jq.cases.cnot3()
    params.T, params.nsteps = params.T * 40 / params.nsteps, 40
    pcof = np.array(load_golden("cnot3")["pcof0"])
    assert int((np.diag(params.Hconst) == 0.0).sum()) >= 2
    nodes, weights = np.array([-0.03, 0.011, 0.02]), np.array([0.3, 0.5, 0.2])
    shift = 0.01 * np.arange(params.Ntot)
    sp = Oracle(params).eval_f_g_grad(pcof, nodes, weights, shift)
    de = Oracle(params, use_sparse=False).eval_f_g_grad(pcof, nodes, weights, shift)
    inf, grad = 0.0, np.zeros(pcof.size)
    H = params.Hconst.copy()
    for ep, w in zip(nodes, weights):
        params.Hconst = H + np.diag(ep * shift)
        r = Oracle(params).traceobjgrad(pcof)
        inf += w * r["primaryobjf"]
        grad += w * r["infidelgrad"]
    params.Hconst = H
    for e in (sp, de):
        assert abs(e["last_infidelity"] - inf) < 1e-13 * abs(inf)
        assert np.linalg.norm(e["last_infidelity_grad"] - grad) < 1e-12 * np.linalg.norm(grad)
    # ... and the perturbation of those levels matters at the test tolerances (a dropped shift is not within 1e-10)
    shift0 = np.where(np.diag(H) == 0.0, 0.0, shift)
    assert abs(Oracle(params).eval_f_g_grad(pcof, nodes, weights, shift0)["last_infidelity"] - inf) > 1e-8 * abs(inf)


def test_stormer_verlet_error_matrix_golden():
    """test/test-stormer-verlet.jl:137-172 against reference_solutions/err-mat-ref.jld2 (<= 1e-13)."""
    ref = np.array(load_golden("err-mat")["err_mat"])
    cfls = 10.0 ** np.arange(-1.0, -2.01, -0.5)
    err = np.zeros((3, 2, 4))
    for j in range(4):
        for i, cfl in enumerate(cfls):
            err[i, 0, j], err[i, 1, j] = _timesteptest(cfl, j)
    assert np.max(np.abs(err - ref)) <= 1e-13
