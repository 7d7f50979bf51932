"""GPU (-m gpu), round 5:
 (1) mid-size ensembles: the backward sweep with the state and the adjoint chain of a column quad on two waves, one time step apart
     (k_backward_qsplit, jq_quad_split_kernels.h) -- bit-identical to the one-wave quad-layout kernel it replaces, 1e-13 next to the
     two-round cooperative-quad sweep, 1e-10 against the oracle, golden-pinned at full length."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

TOL = 1e-10


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _with_env(env, fn):
    """fn() with `env` -- options in the historic spelling ({"JQ_QUAD": "0"} = option quad=0) -- as the options of every handle it creates"""
    import juqbox_jl_amd as jqm
    with jqm.options(**env):
        return fn()


def _cnot3(jq, nsteps):
    params, info = jq.cases.cnot3()
    if nsteps:
        params.T, params.nsteps = params.T * nsteps / params.nsteps, nsteps
    pcof = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "cnot3.json")))["pcof0"])
    return params, pcof


def _problem(jq, kind):
    """cnot3 shortened (4 x 4 x 6: single-subsystem controls, the ORD variant, even m) or random 4 x 4 x n problems (generic trace
    products; NT = 3 with an odd m, three controls and objFuncType 3; NT = 2 with two columns per evaluation: two samples per quad)"""
    if kind == "cnot3":
        return _cnot3(jq, 901)
    from test_gpu_random import random_problem
    Ntot, N, Nc, m, oft = {"t4x3": (48, 4, 3, 5, 3), "t4x2": (32, 2, 2, 4, 1), "t4x5": (80, 4, 4, 3, 2)}[kind]
    rng = np.random.default_rng(50 + Ntot)
    return random_problem(jq, rng, Ntot, N, Nc, 2, 57, m, oft, "t4")


def _eval(jq, params, pcof, nodes, weights, shift, env):
    def run():
        wa = jq.Working_Arrays_HIP(params, pcof.size)
        jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
        t = wa.last_timing()
        out = (params.last_infidelity, params.last_leak, params.last_infidelity_grad.copy(),
               params.last_leak_grad.copy() if params.objFuncType != 1 else np.zeros(1), t)
        wa.close()
        return out
    return _with_env(env, run)


@pytest.mark.parametrize("kind,nsamples,chunk", [("cnot3", 513, 250), ("cnot3", 1024, 0), ("cnot3", 999, 333), ("t4x3", 600, 20), ("t4x3", 1001, 7),
                                                 ("t4x2", 1500, 19), ("t4x5", 700, 57)])
def test_split_backward_sweep_is_the_one_wave_kernel_bit_for_bit(jq, kind, nsamples, chunk):
    """k_backward_qsplit<NT, ORD, 4>: each chain performs the operations of k_backward<NT, 7, 1> in the same order, the trace sums of a
    workgroup are added in the same wave order -- bit-identical results: full and ragged ensembles (idle quads in the last slab, fewer
    slabs than CUs), several chunks with odd lengths (the pipeline fills and drains per chunk, n + 1 super-steps for n steps; chunks
    of one step), odd / even numbers of Neumann terms, NT = 2, 3, 5, 6, two samples per column quad (per-lane shifts and weights),
    four controls, objFuncType 2 / 3 (a second, unforced backward pass), and run to run."""
    params, pcof = _problem(jq, kind)
    rng = np.random.default_rng(nsamples)
    nodes, weights = 0.02 * rng.standard_normal(nsamples), rng.random(nsamples)
    shift = 0.01 * np.arange(params.Ntot)
    env = {"JQ_CHUNK_STEPS": str(chunk)} if chunk else {}
    a = _eval(jq, params, pcof, nodes, weights, shift, env)
    b = _eval(jq, params, pcof, nodes, weights, shift, dict(env, JQ_QSPLIT="0"))
    c = _eval(jq, params, pcof, nodes, weights, shift, env)
    assert a[4]["kernel_family"] == 6 and a[4]["kernel_variant"] == 24 and b[4]["kernel_family"] == 6 and b[4]["kernel_variant"] == 0, (a[4], b[4])
    for x in (b, c):
        assert a[0] == x[0] and a[1] == x[1] and np.array_equal(a[2], x[2]) and np.array_equal(a[3], x[3])
    if kind == "cnot3":
        # three single-subsystem controls: the variant whose twelve trace products ride along in the passes of the adjoint step
        # (k_backward_qsplit<.., RIDE>; the default with two quads per workgroup, forced here with JQ_QS_RIDE=1).  tr5 is then the sum of
        # two dot products instead of a dot product with a sum: the only difference in rounding (observed 4e-17); bit-stable run to run
        env["JQ_QS_RIDE"] = "1"
        d = _eval(jq, params, pcof, nodes, weights, shift, env)
        e = _eval(jq, params, pcof, nodes, weights, shift, env)
        assert d[4]["kernel_variant"] == 24
        assert abs(d[0] - a[0]) <= 1e-13 * abs(a[0]) and abs(d[1] - a[1]) <= 1e-13 * abs(a[1]) and rel(d[2], a[2]) <= 1e-13
        assert d[0] == e[0] and d[1] == e[1] and np.array_equal(d[2], e[2])


@pytest.mark.parametrize("kind,nsamples,mode", [("cnot3", 7, "qw4"), ("cnot3", 7, "qw2"), ("t4x3", 5, "qw4"), ("t4x3", 6, "qw2"), ("t4x2", 11, "qw4"),
                                                ("t4x5", 3, "qw2")])
def test_split_backward_sweep_against_the_oracle(jq, kind, nsamples, mode):
    """Small ensembles forced onto the split kernels (JQ_CQ=0: the quad-layout plan, four quads per workgroup; JQ_QSPLIT=2 JQ_CQ3=0: the
    cooperative-quad plan with the backward sweep on two quads per workgroup) against the oracle's loop over the samples
    (src/ipopt_interface.jl:38-65): infidelity, leak and both gradients at 1e-10."""
    from oracle.oracle import Oracle
    params, pcof = _problem(jq, kind)
    if kind == "cnot3":
        params.T, params.nsteps = params.T * 300 / params.nsteps, 300
    rng = np.random.default_rng(7 + nsamples)
    nodes, weights = 0.02 * rng.standard_normal(nsamples), rng.random(nsamples)
    shift = 0.01 * np.arange(params.Ntot)
    ref = Oracle(params).eval_f_g_grad(pcof, nodes, weights, shift)
    env = {"JQ_CQ": "0", "JQ_CHUNK_STEPS": "41"} if mode == "qw4" else {"JQ_QSPLIT": "2", "JQ_CQ3": "0", "JQ_CHUNK_STEPS": "41"}
    a = _eval(jq, params, pcof, nodes, weights, shift, env)
    assert (a[4]["kernel_family"], a[4]["kernel_variant"]) == ((6, 24) if mode == "qw4" else (8, 22)), a[4]
    gn = np.linalg.norm(ref["last_infidelity_grad"])
    assert abs(a[0] - ref["last_infidelity"]) <= TOL * abs(ref["last_infidelity"])
    assert abs(a[1] - ref["last_leak"]) <= TOL * abs(ref["last_leak"])
    assert np.linalg.norm(a[2] - ref["last_infidelity_grad"]) <= TOL * gn
    if params.objFuncType != 1:
        assert np.linalg.norm(a[3] - ref["last_leak_grad"]) <= TOL * gn


@pytest.mark.parametrize("nsamples", [300, 512])
def test_two_quads_per_workgroup_next_to_the_two_round_cooperative_quad_sweep(jq, nsamples):
    """More column quads than CUs on the cooperative-quad plan: the forward sweep stays on k_forward_cq (two quads per workgroup), the
    backward sweep moves from two rounds of k_backward_cq to k_backward_qsplit<.., 2> (half a slab per workgroup, one wave per SIMD).
    Different kernels, the same arithmetic per column: the trace sums are grouped differently (1e-13); bit-stable run to run."""
    params, pcof = _cnot3(jq, 901)
    nodes, weights, shift = jq.cases.cnot3_ensemble(nsamples)
    a = _eval(jq, params, pcof, nodes, weights, shift, {"JQ_CHUNK_STEPS": "250"})
    b = _eval(jq, params, pcof, nodes, weights, shift, {"JQ_CHUNK_STEPS": "250", "JQ_QSPLIT": "0"})
    c = _eval(jq, params, pcof, nodes, weights, shift, {"JQ_CHUNK_STEPS": "250"})
    assert (a[4]["kernel_family"], a[4]["kernel_variant"]) == (8, 22) and (b[4]["kernel_family"], b[4]["kernel_variant"]) == (8, 0), (a[4], b[4])
    assert abs(a[0] - b[0]) <= 1e-13 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-13 * abs(b[1]) and rel(a[2], b[2]) <= 1e-13
    assert a[0] == c[0] and a[1] == c[1] and np.array_equal(a[2], c[2])


def test_split_kernels_reproduce_the_cnot3_golden_at_full_length(jq):
    """1 024 and 512 unperturbed samples with weights summing to one at the reference's full length (32 386 steps) ARE the reference's
    golden evaluation (test/reference_solutions/cnot3-ref.jld2): infidelity / leak decomposition and gradient (golden minus its
    Tikhonov part) at the reference's tolerance -- on the kernels mid-size ensembles now take (as
    tests/test_gpu_parity.py::test_bench_workload_kernels_reproduce_the_cnot3_golden_at_full_size does for the benchmark's)."""
    from conftest import case_inputs, reference_pass
    params, info, pcof, golden = case_inputs("cnot3")
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    rng = np.random.default_rng(5)
    for ns, variant in ((1024, (6, 24)), (512, (8, 22))):
        w = rng.random(ns)
        w /= w.sum()
        jq.eval_f_g_grad(pcof, params, wa, np.zeros(ns), w, True)
        t = wa.last_timing()
        assert (t["kernel_family"], t["kernel_variant"]) == variant, t
        assert abs(params.last_infidelity - 0.9181500713381303) < 1e-12
        assert abs(params.last_leak - 2.8775930168455916e-05) < 1e-15
        g = np.array(golden["grad0"]) - jq.setup_utils.tikhonov_grad(pcof, params.tik0)
        gt = params.last_infidelity_grad + (params.last_leak_grad if params.objFuncType != 1 else 0.0)
        assert reference_pass(gt, g), ns
    wa.close()


def test_split_kernel_perturbed_samples_at_full_length_match_the_oracle(jq):
    """The path mid_size_ensembles times -- cnot3 at full length, 1 024 PERTURBED samples -- against the oracle: one-hot weights select
    the first and the last sample (different workgroups, waves and quads)."""
    from test_gpu_round2 import oracle_sample
    params, pcof = _cnot3(jq, 0)
    ns = 1024
    nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    for i in (0, 1023):
        r = oracle_sample(params, pcof, nodes[i], shift)
        w = np.zeros(ns)
        w[i] = 1.0
        jq.eval_f_g_grad(pcof, params, wa, nodes, w, True, shift=shift)
        t = wa.last_timing()
        assert (t["kernel_family"], t["kernel_variant"]) == (6, 24), t
        assert abs(params.last_infidelity - r["primaryobjf"]) <= TOL * abs(r["primaryobjf"]), i
        assert abs(params.last_leak - r["secondaryobjf"]) <= TOL * abs(r["secondaryobjf"]), i
        assert rel(params.last_infidelity_grad, r["totalgrad"]) < TOL, i
    wa.close()


# ---- (2) the three-workgroup latency kernels check the co-residency they need -------------------------------------------------------

def test_two_handles_in_two_threads_in_the_split_regime(jq):
    """verdict, round 4: k_backward_cq3's workgroups wait for each other, HIP promises neither their placement nor their co-residency,
    and two handles launching split grids on one GPU at the same time is where co-residency really breaks (every quad could burn its
    ~ 2 s timeout).  Round 5: inside a process the split takes the device EXCLUSIVELY (DevGate) -- an evaluation takes it only when
    no other evaluation of the process is in flight there, and others wait until it is through.  Two threads, each with its own
    handle, 12 evaluations each: every result bit-identical to the single-threaded one, no fault recorded, and the wall time bounded by
    the serial time of all 24 evaluations (+ 50 %: thread start-up, the other thread's forward sweep in front of the gate) -- not by
    timeouts."""
    import threading
    import time
    params, pcof = _cnot3(jq, 2000)
    nodes, weights, shift = jq.cases.cnot3_ensemble(9)
    was = [jq.Working_Arrays_HIP(params, pcof.size) for _ in range(2)]
    ps = [params, _cnot3(jq, 2000)[0]]      # (the mirror keeps the results on the params object: one per thread)
    jq.eval_f_g_grad(pcof, ps[0], was[0], nodes, weights, True, shift=shift)
    assert was[0].last_timing()["kernel_variant"] == 3
    ref = (ps[0].last_infidelity, ps[0].last_leak, ps[0].last_infidelity_grad.copy())
    jq.eval_f_g_grad(pcof, ps[1], was[1], nodes, weights, True, shift=shift)      # (warm-up of the second handle)
    t0 = time.perf_counter()
    for k in range(4):
        jq.eval_f_g_grad(pcof, ps[0], was[0], nodes, weights, True, shift=shift)
    t_one = (time.perf_counter() - t0) / 4
    bad, splits = [], [0, 0]

    def work(i):
        for k in range(12):
            jq.eval_f_g_grad(pcof, ps[i], was[i], nodes, weights, True, shift=shift)
            splits[i] += was[i].last_timing()["kernel_variant"] == 3
            if not (ps[i].last_infidelity == ref[0] and ps[i].last_leak == ref[1] and np.array_equal(ps[i].last_infidelity_grad, ref[2])):
                bad.append((i, k))
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    assert not bad, bad
    for wa in was:
        ls = wa.plan_info()["latency_split"]
        assert ls["faults"] == 0 and ls["off"] is False, ls
        wa.close()
    assert sum(splits) >= 12, splits                     # (whoever finds the device free takes the split)
    assert wall <= 1.5 * 24 * t_one + 0.5, (wall, t_one)


def test_same_device_sub_handles_never_take_the_split(jq):
    """The sub-handles of a same-device multi handle (JQ_MULTI_SAME_DEVICE test mode) evaluate their shards at the same time on one GPU:
    none of them may hold the device exclusively -- decided by the gate now, not by reading the environment variable."""
    params, pcof = _cnot3(jq, 600)
    nodes, weights, shift = jq.cases.cnot3_ensemble(12)
    single = _eval(jq, params, pcof, nodes, weights, shift, {})
    assert single[4]["kernel_variant"] == 3

    def run():
        wam = jq.Working_Arrays_HIP(params, pcof.size, devices=3)
        jq.eval_f_g_grad(pcof, params, wam, nodes, weights, True, shift=shift)
        out = (params.last_infidelity, params.last_infidelity_grad.copy(), wam.plan_info()["latency_split"])
        wam.close()
        return out
    inf, grad, ls = _with_env({"JQ_MULTI_SAME_DEVICE": "1"}, run)
    assert abs(inf - single[0]) <= 1e-13 * abs(single[0]) and rel(grad, single[2]) <= 1e-12
    assert ls["faults"] == 0 and (ls["last_decision"].startswith("not taken: another evaluation") or ls["last_decision"].startswith("taken")), ls


# ---- (3) the dense-operator path (bench.py's dense_operator block) ------------------------------------------------------------------

@pytest.mark.parametrize("nsteps", [6001, pytest.param(0, marks=pytest.mark.slow)], ids=["6001_steps", "full_length"])
def test_dense_drift_matches_the_oracle(jq, nsteps):
    """north_star's "dense (H x state-batch) contraction": cnot3's dimensions with a dense Hermitian drift (cases.cnot3_dense), one
    evaluation on the kernels bench.py's dense_operator block times (k_forward / k_backward<6, 5>: dense 16 x 16 x 4 MFMA tiles, no structure
    exploited) against the oracle's DENSE products (src/StormerVerlet.jl:461-504 dense step!): objective, infidelity / leak split and
    gradient at 1e-10.  6 001 steps (odd: several chunks of odd length) in the default suite; the reference's full length (32 386 steps:
    50 s of CPU oracle) is the slow duplicate."""
    from oracle.oracle import Oracle
    params, info = jq.cases.cnot3_dense()
    if nsteps:
        params.T, params.nsteps = params.T * nsteps / params.nsteps, nsteps
    pcof = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "cnot3.json")))["pcof0"])
    r = Oracle(params, use_sparse=False).traceobjgrad(pcof)
    # (single evaluations and small ensembles: since round 6 the cooperative kernels with their operands straight from HBM / L2 -- six waves
    #  per 16 columns instead of one; option coop_max=0: the slab kernels that serve the benchmark's 4 096 samples)
    for opts, fam in ((None, 1), ({"coop_max": 0}, 0)):
        wa = jq.Working_Arrays_HIP(params, pcof.size, options=opts)
        pi = wa.plan_info()
        assert pi["structure"] == "dense" and pi["block_band"] == 5 and pi["tile_rows"] == 6, pi
        objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, params, wa, False, True)
        t = wa.last_timing()
        assert (t["kernel_family"], t["kernel_size"], t["kernel_band"]) == (fam, 6, 5), t
        wa.close()
        assert abs(objfv - r["objfv"]) <= TOL * abs(r["objfv"])
        assert abs(prim - r["primaryobjf"]) <= TOL * abs(r["primaryobjf"]) and abs(sec - r["secondaryobjf"]) <= TOL * abs(r["secondaryobjf"])
        assert rel(tg, r["totalgrad"]) <= TOL
    # ... and the dense drift is not a rounding-level change of cnot3 (the structured kernels would not notice a dropped perturbation)
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "cnot3.json")))
    assert abs(objfv - g["obj0"]) > 1e-3


# ---- (4) bindings: the order of the pushes --------------------------------------------------------------------------------------------

def test_switching_weights_and_solver_in_one_step(jq):
    """advisor, round 4: sync set the solver before the weights, so a script that went from (full leakage weights, Neumann) to (Diagonal
    weights, Jacobi) in ONE step got JQ_EUNSUPPORTED (the Jacobi solver was refused while the full weights were still on the device).
    Diagonal weights now go in before the solver, full weights after it: both directions work and agree with the oracle; and a vector
    wmat_real next to a full wmat_imag is passed on (not dropped), a Diagonal non-zero wmat_imag refused like the Julia binding does."""
    from oracle.oracle import Oracle
    from test_dense_wmat import forbidden_problem
    p, pcof = forbidden_problem("swap02", 2, 25, True, 1)
    full_r, full_i = p.wmat_real.copy(), p.wmat_imag.copy()
    neumann = p.linear_solver
    wa = jq.Working_Arrays_HIP(p, pcof.size)

    def check():
        r = Oracle(p, use_sparse=False).traceobjgrad(pcof)
        objfv, tg = jq.traceobjgrad(pcof, p, wa, False, True)[:2]
        assert abs(objfv - r["objfv"]) <= TOL * abs(r["objfv"]) and rel(tg, r["totalgrad"]) <= TOL
    check()
    p.wmat_real, p.wmat_imag = np.diag(full_r).copy(), np.zeros(p.Ntot)                       # -> Diagonal + Jacobi in one step
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER, max_iter=60, tol=1e-13, nrhs=p.N)
    check()
    p.wmat_real, p.wmat_imag, p.linear_solver = full_r, full_i, neumann                       # ... and back in one step
    check()
    p.wmat_real = np.diag(full_r).copy()                                                      # a vector next to a FULL wmat_imag: both are passed
    with pytest.raises(Exception) as e:                                                       # (diag(Wr) + i Wi is still Hermitian; the oracle takes matrices)
        p.wmat_imag = np.ones(p.Ntot)                                                         # a non-zero DIAGONAL wmat_imag is refused
        jq.traceobjgrad(pcof, p, wa, False, True)
    assert "Hermitian" in str(e.value)
    p.wmat_imag = full_i
    o1 = jq.traceobjgrad(pcof, p, wa, False, True)[0]
    p.wmat_real = np.diag(np.diag(full_r))                                                    # the same weights written as two full matrices
    o2 = jq.traceobjgrad(pcof, p, wa, False, True)[0]
    assert o1 == o2
    wa.close()


def test_all_reduce_self_check_comparison_runs_in_the_same_device_mode(jq):
    """advisor, round 4: the self-check of the first ncclAllReduce had never executed (the same-device test mode sums on the host) and
    scaled its tolerance by the TOTAL's largest entry although two summation orders differ by roundings of the PARTIAL sums.  The
    comparison is a function now (tolerance 1e-13 x the sum of the devices' largest entries, non-finite results reported as such) and
    JQ_RCCL_SELFCHECK=3 runs it in the test mode: host-order sum against the reverse-order sum of the same packed vectors."""
    from conftest import case_inputs
    params, info, pcof, _ = case_inputs("swap02_rn")
    x, w = np.polynomial.legendre.leggauss(11)
    nodes, weights = x * 0.5 * (2 * np.pi * 2e-2), w * 0.5
    shift = params.shift_weights_reference()

    def run():
        wam = jq.Working_Arrays_HIP(params, pcof.size, devices=4)
        for _ in range(2):
            jq.eval_f_g_grad(pcof, params, wam, nodes, weights, True)
        n = wam.plan_info()["rccl_selfchecks"]
        wam.close()
        return n
    assert _with_env({"JQ_MULTI_SAME_DEVICE": "1", "JQ_RCCL_SELFCHECK": "3"}, run) == 2
    assert _with_env({"JQ_MULTI_SAME_DEVICE": "1"}, run) == 0


# ---- (5) latency path, 81 .. 128 samples: two workgroups per column quad ---------------------------------------------------------------

@pytest.mark.parametrize("kind,nsamples,chunk", [("cnot3", 81, 300), ("cnot3", 128, 0), ("cnot3", 97, 9), ("t4x3", 100, 20), ("t4x5", 90, 19)])
def test_backward_sweep_on_two_workgroups_per_quad_is_the_one_workgroup_kernel(jq, kind, nsamples, chunk):
    """Round 5: with 2 x quads <= CUs < 3 x quads (81 .. 128 cnot3 samples) k_backward_cq3<.., NR = 2> gives a column quad two workgroups --
    state re-integration | adjoint step + ALL trace products (the adjoint path of k_backward_cq with the state waves' share of the
    traces; only u, vi05, vr(t_n) cross the ring).  Every chain and every trace sum performs the operations of k_backward_cq (JQ_CQ3=0)
    in the same order: bit-identical results -- ragged groups of 8 quads, several chunks (down to 9 steps: the shortest the split kernels take), odd / even numbers of
    Neumann terms, NT = 3, 5, 6, single-subsystem and generic trace products, objFuncType 2 / 3, and run to run."""
    params, pcof = _problem(jq, kind)
    rng = np.random.default_rng(nsamples)
    nodes, weights = 0.02 * rng.standard_normal(nsamples), rng.random(nsamples)
    shift = 0.01 * np.arange(params.Ntot)
    env = {"JQ_CHUNK_STEPS": str(chunk)} if chunk else {}
    a = _eval(jq, params, pcof, nodes, weights, shift, env)
    b = _eval(jq, params, pcof, nodes, weights, shift, dict(env, JQ_CQ3="0"))
    c = _eval(jq, params, pcof, nodes, weights, shift, env)
    assert (a[4]["kernel_family"], a[4]["kernel_variant"]) == (8, 2) and (b[4]["kernel_family"], b[4]["kernel_variant"]) == (8, 0), (a[4], b[4])
    for x in (b, c):
        assert a[0] == x[0] and a[1] == x[1] and np.array_equal(a[2], x[2]) and np.array_equal(a[3], x[3])


def test_implicit_midpoint_ensembles_never_take_the_split_kernel(jq):
    """Found by the fuzz slice in round 5: the implicit-midpoint quad-layout path also runs "one slab per workgroup", and the first version
    of the selection handed its backward sweep to the Stormer-Verlet split kernel (relative error 0.75).  A mid-size implicit-midpoint
    ensemble stays on its own kernels (family 7) and agrees with the oracle."""
    from oracle.oracle import Oracle
    params, pcof = _problem(jq, "t4x3")
    params.objFuncType = 1
    params.Integrator_id = jq.Implicit_Midpoint
    params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
    ns = 600
    rng = np.random.default_rng(3)
    nodes, weights = 0.02 * rng.standard_normal(ns), np.zeros(ns)
    weights[[0, 311, 599]] = [0.2, 0.5, 0.3]
    shift = 0.01 * np.arange(params.Ntot)
    wa = jq.Working_Arrays_M_HIP(params, pcof.size)
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
    t = wa.last_timing()
    got = (params.last_infidelity, params.last_infidelity_grad.copy())
    wa.close()
    assert t["kernel_family"] == 7 and t["kernel_variant"] == 0, t
    inf, grad = 0.0, np.zeros(pcof.size)
    H0 = params.Hconst.copy()
    for i in (0, 311, 599):      # (the oracle's ensemble loop is the Stormer-Verlet one: per sample with the perturbed drift)
        params.Hconst = H0 + np.diag(nodes[i] * shift)
        r = Oracle(params, use_sparse=False).traceobjgrad_imr(pcof, 100, 1e-12)
        inf += weights[i] * r["primaryobjf"]
        grad += weights[i] * r["infidelgrad"]
    params.Hconst = H0
    assert abs(got[0] - inf) <= 1e-9 * abs(inf)
    assert rel(got[1], grad) <= 1e-9


# ---- (6) full leakage weights on the latency path (real forbidden states, rank <= 4) ------------------------------------------------------

def _real_forbidden(jq, kind, nforb, seed, oft, cplx=False):
    """`kind` with `nforb` random forbidden states (weights 0.5 .. 1.5), REAL -- wmat_imag = 0 -- unless cplx (src/evalobjgrad.jl:214-232)"""
    params, pcof = _problem(jq, kind)
    rng = np.random.default_rng(seed)
    fs = rng.standard_normal((params.Ntot, nforb)) + (1j * rng.standard_normal((params.Ntot, nforb)) if cplx else 0)
    fs = fs / np.linalg.norm(fs, axis=0)
    fw = 0.5 + rng.random(nforb)
    params.forb_states, params.forb_weights = fs.astype(complex), fw
    W = sum(fw[k] * np.outer(fs[:, k], np.conj(fs[:, k])) for k in range(nforb))
    params.wmat_real = np.asfortranarray(np.real(W).copy())
    params.wmat_imag = np.asfortranarray(np.imag(W).copy()) if cplx else np.zeros_like(params.wmat_real)
    params.objFuncType = oft
    return params, pcof


@pytest.mark.parametrize("kind,nforb,oft,chunk", [("cnot3", 1, 1, 0), ("cnot3", 2, 3, 250), ("cnot3", 4, 2, 9), ("t4x3", 3, 3, 20), ("t4x5", 4, 1, 19), ("t4x2", 2, 1, 0)])
def test_real_full_weights_run_on_the_cooperative_quad_kernels(jq, kind, nforb, oft, chunk):
    """Round 5: a REAL weight matrix of rank <= 4 (real forbidden states) no longer sends a single evaluation or a small ensemble to the
    quad-layout kernels (0.45 s + 57 ms per state at cnot3): the cooperative-quad kernels carry the low-rank terms (CqW: the waves of a
    quad leave partial dots with their publications).  1e-10 against the oracle -- objective, leak, gradients, the infidelity / leak
    split of objFuncType 2 / 3, forward-only, ensembles (ragged quads), several chunks (also of one step), odd / even Neumann terms,
    NT = 2, 3, 5, 6; the same numbers as the quad-layout kernels (JQ_CQ_W=0) to rounding."""
    from test_gpu_dense_wmat import compare
    params, pcof = _real_forbidden(jq, kind, nforb, 40 + nforb, oft)
    env = {"JQ_CHUNK_STEPS": str(chunk)} if chunk else {}

    def run():
        wa = jq.Working_Arrays_HIP(params, pcof.size)
        compare(jq, params, pcof, wa, family=8, ensembles=(3, 9), rng=np.random.default_rng(7))
        t = wa.last_timing()
        why = wa.plan_info()["latency_split"]
        wa.close()
        return t, why
    t, why = _with_env(env, run)
    assert t["kernel_family"] == 8 and t["kernel_variant"] == 3, (t, why)      # (the backward sweep on three workgroups per quad, as with Diagonal weights)
    rng = np.random.default_rng(5)
    nodes, weights = 0.02 * rng.standard_normal(21), rng.random(21)
    shift = 0.01 * np.arange(params.Ntot)
    a = _eval(jq, params, pcof, nodes, weights, shift, env)
    b = _eval(jq, params, pcof, nodes, weights, shift, dict(env, JQ_CQ_W="0"))
    c = _eval(jq, params, pcof, nodes, weights, shift, env)
    assert a[4]["kernel_family"] == 8 and b[4]["kernel_family"] == 6, (a[4], b[4])
    assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-12 * abs(b[1]) and rel(a[2], b[2]) <= 1e-11 and rel(a[3], b[3]) <= 1e-11
    assert a[0] == c[0] and a[1] == c[1] and np.array_equal(a[2], c[2]) and np.array_equal(a[3], c[3])      # (run to run: bit-wise)
    # the one-workgroup backward kernel performs the same operations in the same order
    d = _eval(jq, params, pcof, nodes, weights, shift, dict(env, JQ_CQ3="0"))
    assert a[4]["kernel_variant"] == 3 and d[4]["kernel_variant"] == 0, (a[4], d[4])
    assert a[0] == d[0] and a[1] == d[1] and np.array_equal(a[2], d[2]) and np.array_equal(a[3], d[3])


@pytest.mark.parametrize("kind,nsamples,variant", [("cnot3", 100, 2), ("cnot3", 128, 2), ("t4x3", 100, 2), ("t4x5", 70, 3)])
def test_real_full_weights_on_two_workgroups_per_quad(jq, kind, nsamples, variant):
    """81 .. 128 samples with full (real) weights: two workgroups per quad (three for the 70 samples of the four-control problem),
    single-subsystem and generic trace products; bit-identical to the one-workgroup kernel, and the oracle's numbers on three one-hot
    samples."""
    from oracle.oracle import Oracle
    params, pcof = _real_forbidden(jq, kind, 3, 61, 1)
    rng = np.random.default_rng(nsamples)
    nodes, weights = 0.02 * rng.standard_normal(nsamples), np.zeros(nsamples)
    hot = [0, nsamples // 2, nsamples - 1]
    weights[hot] = [0.3, 0.5, 0.2]
    shift = 0.01 * np.arange(params.Ntot)
    a = _eval(jq, params, pcof, nodes, weights, shift, {})
    b = _eval(jq, params, pcof, nodes, weights, shift, {"JQ_CQ3": "0"})
    assert (a[4]["kernel_family"], a[4]["kernel_variant"]) == (8, variant) and (b[4]["kernel_family"], b[4]["kernel_variant"]) == (8, 0), (a[4], b[4])
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
    inf, leak, grad = 0.0, 0.0, np.zeros(pcof.size)
    H0 = params.Hconst.copy()
    for i in hot:
        params.Hconst = H0 + np.diag(nodes[i] * shift)
        r = Oracle(params, use_sparse=False).traceobjgrad(pcof)
        inf += weights[i] * r["primaryobjf"]
        leak += weights[i] * r["secondaryobjf"]
        grad += weights[i] * r["infidelgrad"]
    params.Hconst = H0
    assert abs(a[0] - inf) <= TOL * abs(inf) and abs(a[1] - leak) <= TOL * abs(leak) and rel(a[2], grad) <= TOL


@pytest.mark.parametrize("kind,nforb,oft,chunk,nsamples", [("cnot3", 1, 1, 0, 9), ("cnot3", 2, 3, 250, 100), ("t4x3", 2, 2, 20, 21), ("t4x5", 1, 3, 9, 90),
                                                            ("t4x2", 2, 1, 0, 5)])
def test_complex_full_weights_run_on_the_split_kernels(jq, kind, nforb, oft, chunk, nsamples):
    """A COMPLEX weight matrix (complex forbidden states) of rank <= 2 fills the four slots with a_0, b_0, a_1, b_1.  Its term
    W_i vr(t_n) of hi1 (src/evalobjgrad.jl:886-888) is needed in the middle of the adjoint step of step n: the backward sweep on two /
    three workgroups per quad has it (the state role is steps ahead), the one-workgroup kernel does not -- without the split (JQ_CQ3=0,
    more than 128 samples) the evaluation runs on the quad-layout kernels as before.  1e-10 against the oracle (objective with the
    penalf2imag cross term, gradients, the objFuncType 2 / 3 split, forward-only, ensembles), the same numbers as the quad-layout
    path to rounding, bit-stable run to run."""
    from test_gpu_dense_wmat import compare
    params, pcof = _real_forbidden(jq, kind, nforb, 70 + nforb, oft, cplx=True)
    env = {"JQ_CHUNK_STEPS": str(chunk)} if chunk else {}

    def run():
        wa = jq.Working_Arrays_HIP(params, pcof.size)
        compare(jq, params, pcof, wa, family=8, ensembles=(3, 9), rng=np.random.default_rng(7))
        t = wa.last_timing()
        wa.close()
        return t
    t = _with_env(env, run)
    assert t["kernel_family"] == 8 and t["kernel_variant"] == 3, t
    rng = np.random.default_rng(5)
    nodes, weights = 0.02 * rng.standard_normal(nsamples), rng.random(nsamples)
    shift = 0.01 * np.arange(params.Ntot)
    a = _eval(jq, params, pcof, nodes, weights, shift, env)
    b = _eval(jq, params, pcof, nodes, weights, shift, dict(env, JQ_CQ3="0"))
    c = _eval(jq, params, pcof, nodes, weights, shift, env)
    assert a[4]["kernel_family"] == 8 and a[4]["kernel_variant"] in (2, 3) and b[4]["kernel_family"] == 6, (a[4], b[4])
    assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-12 * abs(b[1]) and rel(a[2], b[2]) <= 1e-11 and rel(a[3], b[3]) <= 1e-11
    assert a[0] == c[0] and a[1] == c[1] and np.array_equal(a[2], c[2]) and np.array_equal(a[3], c[3])


def test_higher_rank_full_weights_stay_on_the_quad_layout_kernels(jq):
    """Four slots: real rank 5 and complex rank 3 run where they ran; complex weights with more samples than the split kernels take
    likewise; a switch of the weights on a live handle re-routes."""
    from test_gpu_dense_wmat import compare, set_forbidden
    params, pcof = _real_forbidden(jq, "cnot3", 5, 9, 1)
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    compare(jq, params, pcof, wa, family=6)
    set_forbidden(params, np.random.default_rng(2), 3, complex_states=True)
    compare(jq, params, pcof, wa, family=6)
    set_forbidden(params, np.random.default_rng(2), 2, complex_states=True)
    compare(jq, params, pcof, wa, family=8)
    nodes, weights = 0.01 * np.random.default_rng(1).standard_normal(130), np.full(130, 1.0 / 130)
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=0.01 * np.arange(params.Ntot))
    assert wa.last_timing()["kernel_family"] == 6, wa.last_timing()      # (130 quads: no split, a complex W cannot take the one-workgroup kernel)
    set_forbidden(params, np.random.default_rng(3), 2, complex_states=False)
    compare(jq, params, pcof, wa, family=8)
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=0.01 * np.arange(params.Ntot))
    assert wa.last_timing()["kernel_family"] == 8, wa.last_timing()      # (... a real one can)
    # Diagonal weights again
    p0, _ = _problem(jq, "cnot3")
    params.wmat_real, params.wmat_imag = p0.wmat_real, p0.wmat_imag
    compare(jq, params, pcof, wa, family=8)
    jq.traceobjgrad(pcof, params, wa, False, True)
    assert wa.last_timing()["kernel_variant"] == 3, wa.last_timing()
    wa.close()


# ---- (7) short runs with odd / even numbers of steps on the slab kernels (a miscompiled object found in round 5) --------------------------

@pytest.mark.parametrize("structure,env,wts", [(False, {"JQ_COOP_MAX": "0"}, True), ("t4", {"JQ_FORCE_DENSE": "1", "JQ_EMBED": "0"}, True),
                                               (False, {"JQ_COOP_MAX": "0"}, False), (True, {"JQ_COOP_MAX": "0", "JQ_OD": "0"}, True)])
def test_short_runs_with_odd_and_even_numbers_of_steps(jq, structure, env, wts):
    """Round 5: `scripts/fuzz_gpu.py` with FUZZ_FOCUS=wfull_cq drew a drift outside the 4 x 4 x n structure together with full weights --
    the dense 96 x 96 slab kernels with the low-rank terms (object w_6_5) -- and 3 steps: infidelity off by 3e-4, gradient by 0.3.  hipcc
    7.2 miscompiles that object in the VGPR register form (odd chunk lengths with Neumann terms only; rounds 3 and 4 shipped it in the
    default form because the compiler crashed on it then).  It is pinned to the default form (csrc/Makefile, tests/test_abi.py); this
    test runs the slab kernels of that size -- with and without weights, dense / forced-dense / banded -- for 3, 4, 5 and 8 steps and
    0, 3 and 6 Neumann terms against the oracle."""
    from oracle.oracle import Oracle
    from test_gpu_random import random_problem
    for ns in (3, 4, 5, 8):
        for m in (0, 3, 6):
            rng = np.random.default_rng(ns + 10 * m)
            p, pcof = random_problem(jq, rng, 96, 4, 2, 1, ns, m, 3, structure)
            if wts:
                fs = rng.standard_normal((96, 2)) + 1j * rng.standard_normal((96, 2))
                fs = fs / np.linalg.norm(fs, axis=0)
                W = sum((0.5 + 0.3 * k) * np.outer(fs[:, k], np.conj(fs[:, k])) for k in range(2))
                p.wmat_real, p.wmat_imag = np.asfortranarray(W.real.copy()), np.asfortranarray(W.imag.copy())
            wa = _with_env(env, lambda: jq.Working_Arrays_HIP(p, pcof.size))
            r = Oracle(p, use_sparse=False).traceobjgrad(pcof)
            out = jq.traceobjgrad(pcof, p, wa, False, True)
            t = wa.last_timing()
            wa.close()
            assert t["kernel_family"] in (0, 1), t
            gn = np.linalg.norm(r["totalgrad"])
            assert abs(out[2] - r["primaryobjf"]) <= TOL * abs(r["primaryobjf"]), (ns, m, t)
            assert abs(out[3] - r["secondaryobjf"]) <= TOL * abs(r["secondaryobjf"]), (ns, m, t)
            assert np.linalg.norm(out[1] - r["totalgrad"]) <= TOL * gn, (ns, m, t)


# ---- (8) hand-offs of the split kernels when the roles of a quad do NOT start together ---------------------------------------------------

@pytest.mark.parametrize("kind,nsamples,wts,chunk", [("cnot3", 9, None, 300), ("cnot3", 9, "real", 300), ("cnot3", 80, "complex", 300), ("cnot3", 100, "real", 300),
                                                     ("t4x3", 40, "complex", 0), ("t4x5", 90, None, 0), ("cnot3", 128, None, 9), ("cnot3", 33, "complex", 10),
                                                     ("t4x3", 100, "real", 9), ("cnot3", 70, None, 9)])
def test_split_kernels_with_late_roles(jq, kind, nsamples, wts, chunk):
    """Round 5: the first version of the full-weights hand-off left the dots of the chunk's initial state in a ring slot that the state role
    overwrites at step 7 -- and the state role waits for nobody before step 8.  On an idle GPU the roles of a quad start together and every
    test was bit-identical; next to two load processes 43 of 240 weighted evaluations differed (`scripts/soak_cq3_load.py` with
    JQ_SOAK_WEIGHTS=1).  JQ_DEBUG=16 / 32 makes the consumer roles / the state role of every quad start ~ 5 ms late: the results must not
    change by a bit -- with and without weights, three and two workgroups per quad, several chunks down to 9 steps.  (With a FIRST chunk
    of at most 8 steps -- the length of the ring -- the state role can finish the launch, and overwrite the state file the consumers'
    carries start from, before they have started: the hook showed that too; such sweeps take the one-workgroup kernel, next test.)"""
    if wts:
        params, pcof = _real_forbidden(jq, kind, 2, 11, 3, cplx=(wts == "complex"))
    else:
        params, pcof = _problem(jq, kind)
    rng = np.random.default_rng(nsamples)
    nodes, weights = 0.02 * rng.standard_normal(nsamples), rng.random(nsamples)
    shift = 0.01 * np.arange(params.Ntot)
    env = {"JQ_CHUNK_STEPS": str(chunk)} if chunk else {}
    if chunk and chunk < 10 and kind == "cnot3":      # (short chunks: a shorter run, the launches are what is tested)
        params.T, params.nsteps = params.T * 60 / params.nsteps, 60
    a = _eval(jq, params, pcof, nodes, weights, shift, env)
    assert a[4]["kernel_family"] == 8 and a[4]["kernel_variant"] in (2, 3), a[4]
    for bit in ("16", "32"):
        b = _eval(jq, params, pcof, nodes, weights, shift, dict(env, JQ_DEBUG=bit))
        assert b[4]["kernel_variant"] == a[4]["kernel_variant"], (a[4], b[4])
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), bit


@pytest.mark.parametrize("nsamples", [1, 9, 80])
def test_implicit_midpoint_split_kernel_with_late_roles(jq, nsamples):
    """... and the three-workgroup kernel of the implicit-midpoint path (k_backward_cq_imr3, the same hand-off ring)."""
    params, pcof = _cnot3(jq, 901)
    params.Integrator_id = jq.Implicit_Midpoint
    params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
    nodes, weights, shift = jq.cases.cnot3_ensemble(nsamples)

    def run(env):
        def go():
            wa = jq.Working_Arrays_M_HIP(params, pcof.size)
            jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
            t = wa.last_timing()
            out = (params.last_infidelity, params.last_leak, params.last_infidelity_grad.copy(), t)
            wa.close()
            return out
        return _with_env(env, go)
    a = run({"JQ_CHUNK_STEPS": "300"})
    assert a[3]["kernel_family"] == 9 and a[3]["kernel_variant"] == 3, a[3]
    for bit in ("16", "32"):
        b = run({"JQ_CHUNK_STEPS": "300", "JQ_DEBUG": bit})
        assert b[3]["kernel_variant"] == 3
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]), bit


@pytest.mark.parametrize("chunk,nsteps", [(1, 60), (8, 60), (0, 7)])
def test_short_first_chunks_take_the_one_workgroup_kernel(jq, chunk, nsteps):
    """The split kernels' consumer roles read the sweep's initial state from the state file, which the state role overwrites when it is
    through with the chunk; it has to wait for them only from step 8 on.  A first chunk of <= 8 steps (tests with tiny chunks, problems
    with a handful of steps) is not given to them -- `plan_info` says why -- and the late-start hook changes nothing."""
    params, pcof = _cnot3(jq, nsteps)
    nodes, weights, shift = jq.cases.cnot3_ensemble(9)
    env = {"JQ_CHUNK_STEPS": str(chunk)} if chunk else {}

    def run(extra):
        def go():
            wa = jq.Working_Arrays_HIP(params, pcof.size)
            jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
            out = (params.last_infidelity, params.last_leak, params.last_infidelity_grad.copy(), wa.last_timing(), wa.plan_info()["latency_split"]["last_decision"])
            wa.close()
            return out
        return _with_env(dict(env, **extra), go)
    a = run({})
    assert a[3]["kernel_family"] == 8 and a[3]["kernel_variant"] == 0 and "first chunk" in a[4], (a[3], a[4])
    b = run({"JQ_DEBUG": "16"})
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
    c = run({"JQ_CHUNK_STEPS": "9"}) if nsteps > 9 else None
    if c is not None:
        assert c[3]["kernel_variant"] == 3, c[3]
