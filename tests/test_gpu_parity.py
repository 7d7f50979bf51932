"""GPU (-m gpu): parity of the HIP path, called through the C ABI (libjuqbox_hip.so), against
 (1) the reference's golden vectors at the reference's tolerance (rtol 1e-10 / atol 1e-14),
 (2) the CPU oracle on identical inputs (objective, gradient, per-step states, ensembles),
 (3) size-independent properties at the full BASELINE sizes (linearity of the ensemble, unitarity,
     adjoint-vs-finite-difference consistency).
Floating-point tolerance for GPU-vs-oracle comparisons: 1e-10 relative (north_star: "within a stated
fp64 tolerance"); observed differences are ~1e-14."""
import numpy as np
import pytest
from conftest import case_inputs, reference_pass

pytestmark = pytest.mark.gpu

TOL = 1e-10


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def hip(jq):
    from juqbox_jl_amd import _lib
    L = _lib.load()
    assert L.jq_device_count() >= 1, "no HIP device: the hot path has no CPU fallback"
    return jq


def gpu_eval_like_evalGrad(jq, params, wa, pcof):
    """test/evalGrad.jl:14-25 on top of the mirrored callbacks (Tikhonov included)."""
    objv = jq.eval_f_par(pcof, params, wa, [0.0], [1.0])
    grad = np.zeros(pcof.size)
    jq.eval_grad_f_par(pcof, grad, params, wa, [0.0], [1.0])
    if params.objFuncType != 1:
        leak_grad = np.zeros(pcof.size)
        jq.eval_jac_g_par(pcof, [], [], leak_grad, params, wa)
        return np.array([objv, params.last_leak]), np.concatenate([grad, leak_grad])
    return np.array([objv]), grad


@pytest.mark.parametrize("case", ["rabi", "swap02", "flux", "cnot2", "cnot2-leakieq", "cnot2-jacobi", "cnot3", "cnot3:coop", "cnot3:slab", "cnot3:slab-band"])
def test_reference_golden_through_the_callbacks(hip, case):
    """All seven Stormer-Verlet goldens of the reference (test/runtests.jl:30).  cnot3 runs four times: on the quad-layout
    kernels (default for a small batch of this structure), with JQ_QUAD=0 on the cooperative kernels, with JQ_COOP_MAX=0 (and
    JQ_QUAD=0) on the slab kernels (JQ_BW_T4 variant: v_mfma_f64_4x4x4 for the 4x4 diagonal blocks, DPP FMAs for the diagonal
    couplings) and with JQ_OD=0 on the plain block-band slab kernels."""
    import os
    jq = hip
    case, _, mode = case.partition(":")
    params, info, pcof, golden = case_inputs(case)
    opts = {}
    if mode:
        opts["quad"] = 0
    if mode.startswith("slab"):
        opts["coop_max"] = 0
    if mode == "slab-band":
        opts["od"] = 0
    wa = jq.Working_Arrays_HIP(params, pcof.size, options=opts)
    obj, grad = gpu_eval_like_evalGrad(jq, params, wa, pcof)
    assert reference_pass(obj, golden["obj0"]), (obj, golden["obj0"])
    assert reference_pass(grad, golden["grad0"])
    wa.close()


@pytest.mark.parametrize("case", ["rabi", "swap02", "flux", "cnot1", "cnot2", "cnot2-leakieq"])
def test_traceobjgrad_matches_oracle(hip, case):
    from oracle.oracle import Oracle
    jq = hip
    params, info, pcof, _ = case_inputs(case)
    r = Oracle(params).traceobjgrad(pcof)
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, params, wa, False, True)
    # abs 1e-14 or rel 1e-10, the reference's own criterion (test/evalGrad.jl:56-60); rabi's infidelity
    # is 6e-7 = 1 - |s|^2 with |s|^2 ~ 1, so its absolute floor is the ulp of 1
    assert abs(objfv - r["objfv"]) <= max(TOL * abs(r["objfv"]), 1e-14)
    assert abs(prim - r["primaryobjf"]) <= max(TOL * abs(r["primaryobjf"]), 1e-14)
    assert abs(sec - r["secondaryobjf"]) <= max(TOL * abs(r["secondaryobjf"]), 1e-18)
    assert abs(tinf - r["traceInfidelity"]) <= max(TOL * abs(r["traceInfidelity"]), 1e-14)
    assert rel(tg, r["totalgrad"]) < TOL
    assert rel(ig, r["infidelgrad"]) < TOL
    if params.objFuncType != 1:
        assert np.linalg.norm(lg - r["leakgrad"]) < TOL * np.linalg.norm(r["totalgrad"])
    else:
        assert lg.size == 0                       # the reference returns zeros(0) (src/evalobjgrad.jl:808)
    # forward-only return tuple (:1035)
    o2, p2, s2 = jq.traceobjgrad(pcof, params, wa, False, False)
    assert (o2, p2, s2) == (objfv, prim, sec)
    wa.close()


@pytest.mark.parametrize("case", ["swap02", "cnot2"])
def test_state_history_matches_oracle_per_step(hip, case):
    """verbose path: usaver + im*usavei, [Ntot,N,nsteps+1] (src/evalobjgrad.jl:677-680, :748-752, :1031)"""
    from oracle.oracle import Oracle
    jq = hip
    params, info, pcof, _ = case_inputs(case)
    r = Oracle(params).traceobjgrad(pcof, evaladjoint=False, history=True)
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    objfv, hist, fid = jq.traceobjgrad(pcof, params, wa, True, False)
    assert hist.shape == (params.Ntot, params.N, params.nsteps + 1)
    assert np.max(np.abs(hist - r["history"])) < 1e-11
    assert abs(objfv - r["objfv"]) <= TOL * abs(r["objfv"])
    assert abs(fid - (1.0 - r["traceInfidelity"])) < 1e-12
    # column norms stay 1 up to the O(h^2) defect of the (symplectic, not norm-preserving) scheme
    # (the reference prints the same defect in its verbose diagnostics, :980-988)
    nrm = np.sqrt(np.sum(np.abs(hist) ** 2, axis=0))
    assert np.max(np.abs(nrm - 1.0)) < 1e-3
    assert np.max(np.abs(nrm - np.sqrt(np.sum(np.abs(r["history"]) ** 2, axis=0)))) < 1e-12
    wa.close()


@pytest.mark.parametrize("case,nquad", [("swap02", 7), ("cnot2-leakieq", 3), ("swap02_rn", 37)])
def test_ensemble_matches_oracle_loop(hip, case, nquad):
    """eval_f_g_grad! with the reference's Hconst perturbation (src/ipopt_interface.jl:38-65); ragged
    ensembles (nquad not a multiple of the samples per slab) included."""
    from oracle.oracle import Oracle
    jq = hip
    params, info, pcof, _ = case_inputs(case)
    x, w = np.polynomial.legendre.leggauss(nquad)
    nodes, weights = x * 0.5 * (2 * np.pi * 2e-2), w * 0.5
    # the reference's factor 0.01*10^(j-2) is only usable for tiny Ntot (it reaches 1e8 at Ntot=12):
    # larger systems use an explicit per-level shift vector
    shift = params.shift_weights_reference() if params.Ntot <= 4 else 0.05 * np.arange(params.Ntot)
    ref = Oracle(params).eval_f_g_grad(pcof, nodes, weights, shift)
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift if params.Ntot > 4 else None)
    assert abs(params.last_infidelity - ref["last_infidelity"]) <= TOL * abs(ref["last_infidelity"])
    assert abs(params.last_leak - ref["last_leak"]) <= TOL * abs(ref["last_leak"])
    assert rel(params.last_infidelity_grad, ref["last_infidelity_grad"]) < TOL
    if params.objFuncType != 1:
        assert np.linalg.norm(params.last_leak_grad - ref["last_leak_grad"]) < TOL * np.linalg.norm(ref["last_infidelity_grad"])
    wa.close()


def test_ensemble_linearity_and_sweep(hip):
    """Property: nquad copies of the unperturbed node with weights summing to 1 reproduce the single
    evaluation; the sweep entry returns per-node objectives equal to single evaluations."""
    jq = hip
    params, info, pcof, _ = case_inputs("cnot2")
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, params, wa, False, True)
    nq = 21
    jq.eval_f_g_grad(pcof, params, wa, np.zeros(nq), np.full(nq, 1.0 / nq), True)
    assert abs(params.last_infidelity - prim) <= 1e-13
    assert abs(params.last_leak - sec) <= 1e-15
    assert rel(params.last_infidelity_grad, tg) < 1e-12
    shift = np.arange(params.Ntot, dtype=np.float64) * 1e-3
    eps = np.array([-0.2, 0.0, 0.1])
    sw = jq.traceobj_sweep(pcof, params, wa, eps, shift)
    assert abs(sw[1, 0] - objfv) <= 1e-13
    H = params.Hconst.copy()
    params.Hconst[np.diag_indices(params.Ntot)] += eps[2] * shift      # scripts mutate params.Hconst
    o2, p2, s2 = jq.traceobjgrad(pcof, params, wa, False, False)
    params.Hconst[:] = H
    assert abs(sw[2, 0] - o2) <= 1e-12 and abs(sw[2, 1] - p2) <= 1e-12
    wa.close()


def test_params_mutations_are_picked_up(hip):
    """Scripts mutate params after construction (wmat_real, max_iter, Utarget): the device follows."""
    from oracle.oracle import Oracle
    jq = hip
    params, info, pcof, _ = case_inputs("cnot2")
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    params.wmat_real = jq.setup_utils.wmatsetup(params.Ne, params.Ng)
    params.linear_solver.max_iter = 3
    params.Utarget_r, params.Utarget_i = params.Utarget_i.copy(), -params.Utarget_r.copy()
    r = Oracle(params).traceobjgrad(pcof)
    objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, params, wa, False, True)
    assert abs(objfv - r["objfv"]) <= TOL * abs(r["objfv"]) and rel(tg, r["totalgrad"]) < TOL
    wa.close()


def test_error_paths(hip):
    from juqbox_jl_amd import _lib
    jq = hip
    params, info, pcof, _ = case_inputs("swap02")
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    with pytest.raises(_lib.JuqboxHipError) as e:        # src/evalobjgrad.jl:604-606
        jq.traceobjgrad(pcof[:3], params, wa)
    assert e.value.code == _lib.JQ_EINVAL
    with pytest.raises(_lib.JuqboxHipError) as e:        # bcparams DimensionMismatch (src/bsplines.jl:178-181)
        jq.traceobjgrad(np.concatenate([pcof, pcof[:2]]), params, wa)
    assert e.value.code == _lib.JQ_EDIM
    wa.close()


def test_jacobi_solver_matches_oracle_and_switches_at_run_time(hip):
    """JACOBI_SOLVER (src/linear_solvers.jl:110-153) incl. its early exit, and switching the solver of an
    existing params object (scripts replace params.linear_solver after construction)."""
    from oracle.oracle import Oracle
    jq = hip
    params, info, pcof, _ = case_inputs("cnot2-jacobi")
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    for max_iter, tol in ((5, 2e-15), (50, 1e-9), (3, 1e-30), (50, 1e-6), (50, 1e-4)):
        params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER, max_iter=max_iter, tol=tol, nrhs=1)
        r = Oracle(params).traceobjgrad(pcof)
        objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, params, wa, False, True)
        # convergence is tested per sample like the reference's jacobi! (round 3): 1e-10 irrespective of tol
        assert abs(objfv - r["objfv"]) <= TOL * abs(r["objfv"])
        assert rel(tg, r["totalgrad"]) < TOL
        # an ensemble whose samples share a 16-column slab (4 columns each) and stop after DIFFERENT iteration counts (loose
        # tolerances): each sample must stop exactly where the reference's serial loop stops it
        rng = np.random.default_rng(max_iter)
        nodes, weights = 0.3 * rng.standard_normal(7), rng.random(7)
        shift = 0.5 * rng.standard_normal(params.Ntot)
        shift[0] = 0.0
        ref = Oracle(params).eval_f_g_grad(pcof, nodes, weights, shift)
        jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
        assert abs(params.last_infidelity - ref["last_infidelity"]) <= TOL * abs(ref["last_infidelity"])
        assert rel(params.last_infidelity_grad, ref["last_infidelity_grad"]) < TOL
    params.linear_solver = jq.lsolver_object(solver=jq.NEUMANN_SOLVER, max_iter=5)
    r = Oracle(params).traceobjgrad(pcof)
    objfv, tg, *_ = jq.traceobjgrad(pcof, params, wa, False, True)
    assert abs(objfv - r["objfv"]) <= TOL * abs(r["objfv"]) and rel(tg, r["totalgrad"]) < TOL
    wa.close()


def test_cnot3_full_size_properties(hip):
    """BASELINE full size (Ntot=96, 32386 steps): forward-only ensemble with several samples per wave
    and more than one wave -- unperturbed nodes must reproduce the golden decomposition, and the
    adjoint gradient must agree with a central finite difference of the objective along a direction."""
    jq = hip
    params, info, pcof, golden = case_inputs("cnot3")
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    shift = np.kron(np.arange(6), np.ones(16)) + np.tile(np.kron(np.arange(4), np.ones(4)), 6) + np.tile(np.arange(4), 24)
    rng = np.random.default_rng(7)
    d = rng.standard_normal(pcof.size)
    d /= np.linalg.norm(d)
    hfd = 1e-6
    # one batch: 9 samples = 3 slabs: eps = 0 (x5), and +-: perturbed pcof is a different pcof, so run 3 calls
    sw0 = jq.traceobj_sweep(pcof, params, wa, np.zeros(9), shift)
    assert np.max(np.abs(sw0[:, 1] - 0.9181500713381303)) < 1e-12
    assert np.max(np.abs(sw0[:, 2] - 2.8775930168455916e-05)) < 1e-15
    op = jq.traceobj_sweep(pcof + hfd * d, params, wa, np.zeros(1), shift)[0, 0]
    om = jq.traceobj_sweep(pcof - hfd * d, params, wa, np.zeros(1), shift)[0, 0]
    g = np.array(golden["grad0"]) - jq.setup_utils.tikhonov_grad(pcof, params.tik0)
    fd = (op - om) / (2 * hfd)
    assert abs(fd - np.dot(g, d)) < 1e-5 * np.linalg.norm(g)
    wa.close()


@pytest.mark.parametrize("nsamples", [700, 1500, 3072, 5000])
def test_planned_kernels_match_the_slab_kernels_on_large_ensembles(hip, nsamples):
    """jq_eval_f_g_grad picks the kernel variant by batch size (quad layout with 1 / 2 / 3 slabs per workgroup, several
    rounds, or the slab kernels); whatever it picks must agree with the slab kernels (JQ_QUAD=0 JQ_COOP_MAX=0), which the
    goldens and the oracle pin, to rounding (cnot3 shortened to 200 steps)."""
    import os
    jq = hip
    params, info, pcof, _ = case_inputs("cnot3")
    params.nsteps = 200
    params.T = params.T * 200 / 32386
    nodes, weights, shift = jq.cases.cnot3_ensemble(nsamples)
    out = {}
    for tag, env in (("plan", {}), ("slab", {"JQ_QUAD": "0", "JQ_COOP_MAX": "0"})):
        with jq.options(**env):
            wa = jq.Working_Arrays_HIP(params, pcof.size)
        jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
        out[tag] = (params.last_infidelity, params.last_leak, params.last_infidelity_grad.copy(), wa.last_timing()["kernel_family"])
        wa.close()
    a, b = out["plan"], out["slab"]
    assert b[3] == 0 and a[3] == 6
    assert abs(a[0] - b[0]) <= 1e-13 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-12 * abs(b[1])
    assert np.linalg.norm(a[2] - b[2]) <= 1e-12 * np.linalg.norm(b[2])


def test_bench_workload_kernels_reproduce_the_cnot3_golden_at_full_size(hip):
    """The kernels bench.py times (3072 samples per GPU: quad layout, three slabs per workgroup) at BASELINE's full size
    (Ntot = 96, 32386 steps, forward + adjoint): with every node unperturbed and weights summing to one the ensemble IS the
    reference's golden evaluation -- infidelity / leak decomposition and gradient (golden minus its
    Tikhonov part) at the reference's tolerance."""
    jq = hip
    params, info, pcof, golden = case_inputs("cnot3")
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    ns = 3072
    rng = np.random.default_rng(5)
    w = rng.random(ns)
    w /= w.sum()
    jq.eval_f_g_grad(pcof, params, wa, np.zeros(ns), w, True)
    t = wa.last_timing()
    assert t["kernel_family"] == 6 and t["kernel_band"] == 7
    assert abs(params.last_infidelity - 0.9181500713381303) < 1e-12
    assert abs(params.last_leak - 2.8775930168455916e-05) < 1e-15
    g = np.array(golden["grad0"]) - jq.setup_utils.tikhonov_grad(pcof, params.tik0)
    gt = params.last_infidelity_grad + (params.last_leak_grad if params.objFuncType != 1 else 0.0)
    assert reference_pass(gt, g)
    wa.close()


def test_plain_c_caller_gets_the_numbers_of_the_python_mirror(hip, tmp_path):
    """examples/c_abi_demo.c (rabi case built by hand in C, no Python in the call path) against the Python mirror on the
    same inputs; the odd-length call returns the code of the reference's error (src/evalobjgrad.jl:604-606)."""
    import subprocess
    from test_abi import build_c_demo
    jq = hip
    exe = build_c_demo(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")
    objfv_c = float(lines[0].split()[1])
    grad_c = np.array([float(ln.split()[2]) for ln in lines if ln.startswith("grad ")])
    assert "wrong_length_rc -1" in r.stdout          # JQ_EINVAL
    params, info = jq.cases.rabi()
    assert params.nsteps == 57 and params.linear_solver.max_iter == 10      # (what the C program hard-codes)
    pcof = info["pcof0"]
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    objfv, tg, *_ = jq.traceobjgrad(pcof, params, wa, False, True)
    assert abs(objfv_c - objfv) <= 1e-13 * max(abs(objfv), 1e-3)
    assert np.max(np.abs(grad_c - tg)) <= 1e-13
    wa.close()
