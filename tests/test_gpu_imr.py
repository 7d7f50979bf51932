"""GPU (-m gpu): the implicit-midpoint path (traceobjgrad for Working_Arrays_M,
src/evalobjgrad.jl:1042-1481) through the C ABI (jq_set_integrator) and the mirrored callbacks."""
import json
import os

import numpy as np
import pytest
from conftest import case_inputs, load_golden, reference_pass

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _imr_params(jq, case):
    params, info, pcof, _ = case_inputs(case)
    params.Integrator_id = jq.Implicit_Midpoint                                    # test/runtests.jl:69-70
    params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
    return params, pcof


@pytest.mark.parametrize("case", ["rabi", "swap02", "flux", "cnot2", "cnot2-leakieq", "cnot2-jacobi", "cnot3", "cnot3:quad", "cnot3:coop"])
def test_reference_imr_golden_through_the_callbacks(jq, case):
    """All seven implicit-midpoint goldens of the reference (second loop of test/runtests.jl:60-80): Ntot <= 16 on the
    row-lane kernels (family 4), cnot3 (Ntot = 96 = 4 x 4 x 6, N = 4) on the cooperative-quad kernels (family 9), with
    JQ_IMR_CQ=0 on the quad-layout kernels (family 7) and, with JQ_QUAD=0, on the cooperative MFMA kernels (family 5)."""
    case, _, mode = case.partition(":")
    params, pcof = _imr_params(jq, case)
    golden = load_golden(case + "-imr")
    env = {"coop": {"JQ_QUAD": "0"}, "quad": {"JQ_IMR_CQ": "0"}}.get(mode, {})
    with jq.options(**env):
        wa = jq.Working_Arrays_M_HIP(params, pcof.size)
        _golden_checks(jq, params, pcof, wa, golden, mode)


def _golden_checks(jq, params, pcof, wa, golden, mode):
    n = pcof.size
    if params.objFuncType == 3:                      # test/evalGrad.jl:14-25
        obj = np.array([jq.eval_f_par(pcof, params, wa), 0.0])
        g = np.zeros(1)
        jq.eval_g_par(pcof, g, params, wa)
        obj[1] = g[0]
        gf, jac = np.zeros(n), np.zeros(n)
        jq.eval_grad_f_par(pcof, gf, params, wa)
        jq.eval_jac_g_par(pcof, [], [], jac, params, wa)
        grad = np.concatenate([gf, jac])
    else:
        obj = np.array([jq.eval_f_par(pcof, params, wa)])
        grad = np.zeros(n)
        jq.eval_grad_f_par(pcof, grad, params, wa)
    assert reference_pass(obj, golden["obj0"]), (obj, golden["obj0"])
    assert reference_pass(grad, golden["grad0"])
    assert wa.last_timing()["kernel_family"] == ({"coop": 5, "quad": 7}.get(mode, 9) if params.Ntot > 16 else 4)
    wa.close()


@pytest.mark.parametrize("case", ["swap02", "cnot1", "cnot2-leakieq"])
def test_imr_matches_oracle_including_history_and_ensemble(jq, case):
    from oracle.oracle import Oracle
    params, pcof = _imr_params(jq, case)
    orc = Oracle(params)
    r = orc.traceobjgrad_imr(pcof, 100, 1e-12, history=True)
    wa = jq.Working_Arrays_M_HIP(params, pcof.size)
    objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, params, wa, False, True)
    gn = np.linalg.norm(r["totalgrad"])
    assert abs(prim - r["primaryobjf"]) <= max(TOL * abs(r["primaryobjf"]), 1e-14)
    assert abs(sec - r["secondaryobjf"]) <= max(TOL * abs(r["secondaryobjf"]), 1e-18)
    assert np.linalg.norm(tg - r["totalgrad"]) <= TOL * gn
    assert np.linalg.norm(ig - r["infidelgrad"]) <= TOL * gn
    if params.objFuncType != 1:
        assert np.linalg.norm(lg - r["leakgrad"]) <= TOL * gn
    _, hist, _ = jq.traceobjgrad(pcof, params, wa, True, False)
    assert np.max(np.abs(hist - r["history"])) < 1e-10
    # weighted ensemble with a diagonal perturbation = weighted sum of single evaluations of the oracle
    rng = np.random.default_rng(3)
    nodes, weights = 0.02 * rng.standard_normal(5), rng.random(5)
    shift = 0.05 * rng.standard_normal(params.Ntot)
    shift[0] = 0.0
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
    inf = leak = 0.0
    g = np.zeros(pcof.size)
    H0 = params.Hconst.copy()
    for ep, wq in zip(nodes, weights):
        params.Hconst = H0 + np.diag(ep * shift)
        rr = Oracle(params).traceobjgrad_imr(pcof, 100, 1e-12)
        inf += wq * rr["primaryobjf"]
        leak += wq * rr["secondaryobjf"]
        g += wq * rr["infidelgrad"]
    params.Hconst = H0
    assert abs(params.last_infidelity - inf) <= TOL * abs(inf)
    assert abs(params.last_leak - leak) <= max(TOL * abs(leak), 1e-18)
    assert np.linalg.norm(params.last_infidelity_grad - g) <= TOL * np.linalg.norm(g)
    wa.close()


def test_imr_with_more_than_16_columns_runs_on_the_parts_kernels(jq):
    """Rounds 1-3 refused more than 16 columns per evaluation (the solver's per-evaluation stopping test needs them in one
    workgroup); round 4: one cooperative workgroup per evaluation walks over its 16-column parts (tests/test_gpu_round4.py)."""
    from oracle.oracle import Oracle
    from test_gpu_random import random_problem
    p, pcof = random_problem(jq, np.random.default_rng(1), 20, 18, 1, 1, 5, 1, 1, False)
    p.Integrator_id = jq.Implicit_Midpoint
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=50, tol=1e-11, nrhs=18)
    p.wmat = p.wmat_real.copy()
    wa = jq.Working_Arrays_M_HIP(p, pcof.size)
    r = Oracle(p, use_sparse=False).traceobjgrad_imr(pcof, 50, 1e-11)
    objfv, tg = jq.traceobjgrad(pcof, p, wa)[:2]
    assert wa.last_timing()["kernel_family"] == 5
    assert abs(objfv - r["objfv"]) <= 1e-9 * abs(r["objfv"]) and np.linalg.norm(tg - r["totalgrad"]) <= 1e-9 * np.linalg.norm(r["totalgrad"])
    wa.close()


@pytest.mark.parametrize("cfg", [(2, 1, 1, 1, 9, 1), (5, 3, 2, 2, 14, 3), (9, 2, 3, 1, 11, 2), (12, 4, 2, 2, 8, 1), (16, 4, 4, 1, 6, 3),
                                 (17, 5, 2, 1, 7, 3), (33, 3, 1, 2, 6, 1), (48, 4, 3, 1, 5, 2), (80, 16, 2, 1, 4, 1), (95, 2, 2, 1, 4, "band"),
                                 (48, 4, 2, 1, 5, "od"), (96, 4, 3, 1, 4, "od"), (32, 1, 2, 1, 6, "t4"), (48, 2, 3, 2, 5, "t4"),
                                 (64, 4, 3, 1, 5, "t4"), (96, 4, 3, 1, 4, "t4"), (80, 3, 2, 1, 4, "t4"), (96, 4, 3, 1, 4, "t4q"),
                                 (48, 4, 4, 2, 7, "t4"), (16, 16, 1, 1, 5, 1), (9, 9, 2, 2, 7, 2), (12, 5, 3, 1, 6, 3), (6, 6, 1, 1, 9, 1),
                                 (16, 7, 2, 1, 6, "t4"), (112, 4, 3, 1, 4, "t4"), (128, 2, 2, 1, 4, "t4"),
                                 # Ntot > 96 without the 4 x 4 x n structure (round 3): cooperative kernels, images from HBM / L2
                                 (100, 3, 2, 1, 4, "band"), (130, 4, 1, 1, 3, 1), (160, 5, 2, 1, 3, "band"), (200, 16, 1, 1, 3, "band"),
                                 (256, 2, 1, 1, 2, 3), (112, 3, 2, 1, 3, "od"),
                                 # dense 96 x 96 (both images of a step do not fit the LDS: the <6, 5> variant that reads them from HBM / L2)
                                 (96, 4, 2, 1, 4, 1), (90, 7, 1, 1, 3, 2),
                                 # 4 x 4 x 7 / 4 x 4 x 8 with N = 3, 5 (no quad-layout kernels for these N: cooperative layout, images from HBM / L2)
                                 (112, 3, 2, 1, 3, "t4"), (128, 5, 1, 1, 3, "t4")],
                         ids=lambda c: "Ntot%d_N%d_Nc%d_f%d_o%s" % (c[0], c[1], c[2], c[3], c[5]))
def test_imr_random_problems_match_oracle(jq, cfg):
    """Sizes and paddings the reference cases do not reach: every row-lane instantiation (NPJ 2..16), N = 1..4 columns
    (one, two or four evaluations per wave; an idle row for N = 3), Ntot <= 16 with 5..16 columns (one-wave cooperative kernels), 1..4 controls, objFuncType 1/2/3, several chunks,
    ensembles with ragged last waves."""
    from oracle.oracle import Oracle
    from test_gpu_random import random_problem
    Ntot, N, Nc, Nfreq, nsteps, oft = cfg
    rng = np.random.default_rng(77 + Ntot)
    structure = False
    if oft == "od":          # Kronecker structure -> the JQ_BW_OD variant of the cooperative kernels
        structure, oft = "od", 1
    env = {"JQ_CHUNK_STEPS": "5"}
    if oft == "t4q":         # ... N = 4 on the quad-layout kernels instead of the cooperative-quad ones
        env["JQ_IMR_CQ"] = "0"
        oft = "t4"
    if oft == "t4":          # 4 x 4 x n Kronecker structure -> N = 4: the cooperative-quad kernels, N = 1, 2: the quad-layout kernels
        structure, oft = "t4", 2             #                                (N = 3: cooperative kernels)
    if oft == "band":        # ladder-operator couplings: block band 1 (dense 96 x 96 images do not fit two LDS slots)
        structure, oft = True, 3
    p, pcof = random_problem(jq, rng, Ntot, N, Nc, Nfreq, nsteps, 3, oft, structure)
    p.Integrator_id = jq.Implicit_Midpoint
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=60, tol=1e-11, nrhs=N)
    p.wmat = p.wmat_real.copy()
    with jq.options(**env):
        wa = jq.Working_Arrays_M_HIP(p, pcof.size)
        _random_checks(jq, p, pcof, wa, rng, Ntot, N, structure, "JQ_IMR_CQ" in env)


def _random_checks(jq, p, pcof, wa, rng, Ntot, N, structure, noncq):
    from oracle.oracle import Oracle
    r = Oracle(p, use_sparse=False).traceobjgrad_imr(pcof, 60, 1e-11, history=True)
    objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, p, wa, False, True)
    gn = np.linalg.norm(r["totalgrad"])
    assert abs(prim - r["primaryobjf"]) <= 1e-9 and abs(sec - r["secondaryobjf"]) <= 1e-9 * max(abs(r["secondaryobjf"]), 1e-3)
    assert np.linalg.norm(tg - r["totalgrad"]) <= 1e-9 * gn and np.linalg.norm(ig - r["infidelgrad"]) <= 1e-9 * gn
    if structure == "t4":
        assert wa.last_timing()["kernel_family"] == (9 if N == 4 and not noncq and Ntot <= 112 else 7 if N in (1, 2, 4) else 5)
    _, hist, _ = jq.traceobjgrad(pcof, p, wa, True, False)
    assert np.max(np.abs(hist - r["history"])) < 1e-10
    for nq in (1, 3, 7):
        nodes, weights = 0.05 * rng.standard_normal(nq), rng.random(nq)
        shift = 0.05 * rng.standard_normal(Ntot)
        shift[0] = 0.0
        jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=shift)
        inf, g, H0 = 0.0, np.zeros(pcof.size), p.Hconst.copy()
        for ep, wq in zip(nodes, weights):
            p.Hconst = H0 + np.diag(ep * shift)
            rr = Oracle(p, use_sparse=False).traceobjgrad_imr(pcof, 60, 1e-11)
            inf += wq * rr["primaryobjf"]
            g += wq * rr["infidelgrad"]
        p.Hconst = H0
        assert abs(p.last_infidelity - inf) <= 1e-9 * abs(inf)
        assert np.linalg.norm(p.last_infidelity_grad - g) <= 1e-9 * np.linalg.norm(g)
    wa.close()


def test_implicit_midpoint_large_batch_on_the_cooperative_kernels(jq):
    """1500 slabs (more than four rounds of one workgroup per CU) of a dense Ntot = 40 problem: one-hot ensemble weights select the
    first, a middle and the last sample, each against the oracle's evaluation of that sample."""
    from oracle.oracle import Oracle
    from test_gpu_random import random_problem

    rng = np.random.default_rng(11)
    Ntot, N = 40, 4
    p, pcof = random_problem(jq, rng, Ntot, N, 2, 1, 8, 3, 1, False)
    p.Integrator_id = jq.Implicit_Midpoint
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=60, tol=1e-11, nrhs=N)
    wa = jq.Working_Arrays_M_HIP(p, pcof.size)
    ns = 6000
    nodes = 0.02 * rng.standard_normal(ns)
    shift = 0.05 * rng.standard_normal(Ntot)
    shift[0] = 0.0
    for j in (0, 2999, 5999):
        w = np.zeros(ns)
        w[j] = 1.0
        jq.eval_f_g_grad(pcof, p, wa, nodes, w, True, shift=shift)
        assert wa.last_timing()["kernel_family"] == 5
        H0 = p.Hconst.copy()
        p.Hconst = H0 + np.diag(nodes[j] * shift)
        try:
            r = Oracle(p, use_sparse=False).traceobjgrad_imr(pcof, 60, 1e-11)
        finally:
            p.Hconst = H0
        assert abs(p.last_infidelity - r["primaryobjf"]) <= 1e-10 * abs(r["primaryobjf"])
        assert np.linalg.norm(p.last_infidelity_grad - r["infidelgrad"]) <= 1e-9 * np.linalg.norm(r["infidelgrad"])
    wa.close()


def test_imr_cooperative_quad_ensemble_matches_quad_layout(jq):
    """45 samples (workgroups on many CUs, a ragged last slab) of a 4 x 4 x 5 problem with objFuncType 2 (two backward passes) and
    three chunks: the cooperative-quad implicit-midpoint kernels against the quad-layout ones -- same fixed-point iterates and
    stopping decisions, only the order of the sums differs."""
    from test_gpu_random import random_problem
    rng = np.random.default_rng(2024)
    p, pcof = random_problem(jq, rng, 80, 4, 3, 2, 13, 3, 2, "t4")
    p.Integrator_id = jq.Implicit_Midpoint
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=60, tol=1e-11, nrhs=4)
    p.wmat = p.wmat_real.copy()
    nodes, weights = 0.05 * rng.standard_normal(45), rng.random(45)
    shift = 0.05 * rng.standard_normal(80)
    shift[0] = 0.0
    res = {}
    for tag, env in (("cq", {}), ("quad", {"JQ_IMR_CQ": "0"})):
        env = dict(env, JQ_CHUNK_STEPS="5")
        with jq.options(**env):
            wa = jq.Working_Arrays_M_HIP(p, pcof.size)
            jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=shift)
            res[tag] = (p.last_infidelity, p.last_leak, p.last_infidelity_grad.copy(), p.last_leak_grad.copy(),
                        wa.last_timing()["kernel_family"])
            wa.close()
    a, b = res["cq"], res["quad"]
    assert a[4] == 9 and b[4] == 7
    assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-12 * abs(b[1])
    assert np.linalg.norm(a[2] - b[2]) <= 1e-11 * np.linalg.norm(b[2])
    assert np.linalg.norm(a[3] - b[3]) <= 1e-11 * np.linalg.norm(b[3])
