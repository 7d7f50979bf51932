"""GPU (-m gpu): round 6 -- what ABI 5 stopped refusing, and the per-handle options.
 (1) Hilbert spaces beyond 256 levels (run-time-size kernels, jq_huge_kernels.h) against the oracle: Neumann and Jacobi solver, ensembles,
     N > 16, full leakage weights, per-step states.  The reference has no size limit (src/evalobjgrad.jl:152-343).
 (2) full leakage weights WITH the Jacobi solver (src/linear_solvers.jl:110-153 accepts any weights) on every plan: cooperative kernels,
     the slab kernels <1, 0> / <6, 5>, 4 x 4 x n plans that are planned again without the structure.
 (3) full weights of rank > 16, more than 16 control Hamiltonians.
 (4) options: per handle, parsed once, nothing read from the environment but JQ_OPTIONS.
 (5) the start-up rendezvous of the split latency kernels: an abandoned launch falls back within the same call."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, reference_pass
from test_gpu_random import random_problem

pytestmark = pytest.mark.gpu


def set_forbidden(p, rng, nforb, complex_states=True):
    fs = rng.standard_normal((p.Ntot, nforb)) + (1j * rng.standard_normal((p.Ntot, nforb)) if complex_states else 0)
    fs = fs / np.linalg.norm(fs, axis=0)
    W = sum((0.5 + rng.random()) * np.outer(fs[:, k], np.conj(fs[:, k])) for k in range(nforb))
    p.wmat_real, p.wmat_imag = np.asfortranarray(W.real.copy()), np.asfortranarray(W.imag.copy())


def check(jq, p, pcof, wa, rng, ensembles=(), history=False):
    """objective, infidelity / leak split, the three gradients [, per-step states, ensembles] against the oracle at the reference's criterion"""
    from oracle.oracle import Oracle
    r = Oracle(p, use_sparse=False).traceobjgrad(pcof, history=history)
    objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, p, wa, False, True)
    assert reference_pass(prim, r["primaryobjf"]) and reference_pass(sec, r["secondaryobjf"]), (prim, r["primaryobjf"], sec, r["secondaryobjf"])
    assert reference_pass(tg, r["totalgrad"]) and reference_pass(ig, r["infidelgrad"])
    if p.objFuncType != 1:
        assert reference_pass(lg, r["leakgrad"])
    if history:
        _, hist, _ = jq.traceobjgrad(pcof, p, wa, True, False)
        assert np.max(np.abs(hist - r["history"])) < 1e-10
    for nq in ensembles:
        nodes, weights = 0.05 * rng.standard_normal(nq), rng.random(nq)
        shift = 0.05 * rng.standard_normal(p.Ntot)
        shift[0] = 0.0
        ref = Oracle(p, use_sparse=False).eval_f_g_grad(pcof, nodes, weights, shift)
        jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=shift)
        assert reference_pass(p.last_infidelity, ref["last_infidelity"]) and reference_pass(p.last_leak, ref["last_leak"])
        assert reference_pass(p.last_infidelity_grad, ref["last_infidelity_grad"])
        if p.objFuncType != 1:
            assert reference_pass(p.last_leak_grad, ref["last_leak_grad"])
    return wa.last_timing()


# ---- (1) beyond 256 levels ------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("cfg", [
    # Ntot, N, Nc, Nfreq, nsteps, m, objFuncType, structure, solver, nforb
    (300, 5, 2, 1, 40, 3, 1, False, "neumann", 0),       # the review's case: dense, 19 tile rows
    (260, 3, 3, 2, 12, 5, 3, True, "neumann", 0),        # banded operators (evaluated as dense windows), two backward passes
    (290, 16, 1, 1, 8, 2, 2, "t4", "jacobi", 0),         # Jacobi solver with the per-sample stopping rule
    (270, 20, 2, 1, 6, 2, 1, False, "neumann", 0),       # N > 16: two slabs per sample
    (272, 4, 2, 1, 9, 4, 3, False, "neumann", 3),        # full (complex) leakage weights
    (257, 2, 5, 1, 7, 1, 1, False, "neumann", 0),        # one row beyond 16 tile rows (ragged last tile), five controls (two groups)
], ids=lambda c: "Ntot%d_N%d_%s_%s_w%d" % (c[0], c[1], c[7] if isinstance(c[7], str) else ("band" if c[7] else "dense"), c[8], c[9]))
def test_hilbert_spaces_beyond_256_levels_match_the_oracle(jq, cfg):
    Ntot, N, Nc, Nfreq, nsteps, m, oft, structure, solver, nforb = cfg
    rng = np.random.default_rng(6000 + Ntot)
    p, pcof = random_problem(jq, rng, Ntot, N, Nc, Nfreq, nsteps, m, oft, structure)
    if solver == "jacobi":
        p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER, max_iter=40, tol=1e-9, nrhs=N)
    if nforb:
        set_forbidden(p, rng, nforb)
    wa = jq.Working_Arrays_HIP(p, pcof.size, options={"chunk_steps": 5} if Ntot == 260 else None)
    plan = wa.plan_info()
    assert plan["tile_rows"] == (Ntot + 15) // 16 > 16 and plan["structure"] == "dense"
    t = check(jq, p, pcof, wa, rng, ensembles=(3,) if N <= 16 else (2,), history=(Ntot == 300))
    assert t["kernel_family"] == 1 and t["kernel_size"] == (Ntot + 15) // 16
    # a drift update keeps the plan (nothing to re-plan at this size)
    p.Hconst = p.Hconst + np.diag(0.01 * rng.standard_normal(Ntot))
    check(jq, p, pcof, wa, rng)
    assert wa.plan_info()["replanned"] is False
    wa.close()


def test_implicit_midpoint_beyond_256_levels_is_refused_not_served_by_something_else(jq):
    from juqbox_jl_amd import _lib
    rng = np.random.default_rng(1)
    p, pcof = random_problem(jq, rng, 260, 2, 1, 1, 4, 2, 1, False)
    p.Integrator_id = jq.Implicit_Midpoint
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=20, tol=1e-10, nrhs=2)
    p.wmat = p.wmat_real.copy()
    wa = jq.Working_Arrays_M_HIP(p, pcof.size)
    with pytest.raises(_lib.JuqboxHipError) as e:
        jq.traceobjgrad(pcof, p, wa, False, True)
    assert e.value.code == _lib.JQ_EUNSUPPORTED and "256" in str(e.value)
    wa.close()


# ---- (2) full weights with the Jacobi solver ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("cfg", [
    # Ntot, N, Nc, nsteps, structure, options, expected (family, size, band), re-planned without the 4 x 4 x n structure
    (48, 5, 2, 9, True, {}, (1, 3, 1), False),                                  # cooperative kernels, block band
    (64, 4, 3, 8, "t4", {}, (1, 4, 9), False),                                  # 4 x 4 x 4 plan: its cooperative (diagonal off-diagonal block) kernels
    (12, 3, 1, 15, False, {}, (0, 1, 0), False),                                # one tile row: slab <1, 0> with both compiled in (x_1_0)
    (16, 4, 2, 11, "t4", {}, (0, 1, 0), True),                                  # 4 x 4 x 1 plan: planned again without the structure
    (112, 4, 2, 6, "t4", {}, (1, 7, 1), True),                                  # 4 x 4 x 7 plan: the general Ntot > 96 cooperative kernels
    (96, 4, 2, 7, False, {"force_dense": 1, "embed": 0, "coop_max": 0}, (0, 6, 5), False),     # dense 96 x 96, large batches (here: forced): slab <6, 5> (x_6_5)
    (96, 4, 2, 7, False, {"force_dense": 1, "embed": 0}, (1, 6, 5), False),     # ... small batches: the cooperative kernels with HBM operands (round 6)
    (200, 6, 1, 5, False, {}, (1, 13, 15), False),                              # Ntot > 96, dense
    (40, 24, 2, 6, False, {}, (1, 3, 2), False),                                # N > 16 on the cooperative kernels
], ids=lambda c: "Ntot%d_%s" % (c[0], c[4] if isinstance(c[4], str) else ("band" if c[4] else "dense")) if isinstance(c, tuple) and len(c) == 8 else None)
def test_full_weights_with_the_jacobi_solver_match_the_oracle(jq, cfg):
    Ntot, N, Nc, nsteps, structure, opts, kernel, replanned = cfg
    rng = np.random.default_rng(6100 + Ntot + N)
    p, pcof = random_problem(jq, rng, Ntot, N, Nc, 1, nsteps, 3, 3, structure)
    # (N > 16 on the cooperative kernels: convergence is tested per 16-column part -- include/juqbox_hip.h -- so the parts may stop at
    #  different iterations and the result carries O(tol); a tolerance at rounding level makes that invisible)
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER, max_iter=80 if N > 16 else 50, tol=1e-13 if N > 16 else 1e-9, nrhs=N)
    set_forbidden(p, rng, 2 + Ntot % 2, complex_states=(Ntot % 3 != 0))
    wa = jq.Working_Arrays_HIP(p, pcof.size, options=opts)
    t = check(jq, p, pcof, wa, rng, ensembles=(2,) if N > 16 else (3, 9))
    assert (t["kernel_family"], t["kernel_size"], t["kernel_band"]) == kernel, t
    plan = wa.plan_info()
    assert (plan["structure"] != "t4") == (replanned or structure != "t4"), plan["structure"]
    # back to the Neumann solver: the structure (and its faster kernels) come back; back to Diagonal weights with Jacobi too
    p.linear_solver = jq.lsolver_object(solver=jq.NEUMANN_SOLVER, max_iter=3)
    check(jq, p, pcof, wa, rng)
    if structure == "t4" and not opts:
        assert wa.plan_info()["structure"] == "t4"
    wa.close()


# ---- (3) ranks beyond 16, more than 16 controls --------------------------------------------------------------------------------------

@pytest.mark.parametrize("cfg", [(40, 5, False, 20, {}, 1), (64, 4, "t4", 18, {}, 6), (96, 3, False, 19, {"force_dense": 1, "embed": 0, "coop_max": 0}, 0), (96, 3, False, 19, {"force_dense": 1, "embed": 0}, 1), (130, 4, True, 24, {}, 1),
                                 (33, 6, False, 33, {}, 1)],
                         ids=lambda c: "Ntot%d_rank%d" % (c[0], c[3]) if isinstance(c, tuple) else None)
def test_full_weights_of_rank_beyond_16(jq, cfg):
    """the reference takes any number of forbidden states (src/evalobjgrad.jl:214-232); rank 33 at Ntot 33 is a FULL-rank weight matrix"""
    Ntot, N, structure, rank, opts, family = cfg
    rng = np.random.default_rng(6200 + Ntot)
    p, pcof = random_problem(jq, rng, Ntot, N, 2, 1, 7, 3, 2, structure)
    set_forbidden(p, rng, rank, complex_states=(Ntot % 2 == 0))
    wa = jq.Working_Arrays_HIP(p, pcof.size, options=opts)
    assert wa.plan_info()["full_weight_rank"] == rank
    t = check(jq, p, pcof, wa, rng, ensembles=(4,))
    assert t["kernel_family"] == family, t
    wa.close()


@pytest.mark.parametrize("Ntot,Nc", [(6, 18), (48, 17)])
def test_more_than_16_control_hamiltonians(jq, Ntot, Nc):
    rng = np.random.default_rng(6300 + Ntot)
    p, pcof = random_problem(jq, rng, Ntot, 3, Nc, 1, 9, 2, 1, False)
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    assert wa.plan_info()["controls"] == Nc and wa.plan_info()["control_groups"] == (Nc + 3) // 4
    check(jq, p, pcof, wa, rng, ensembles=(3,))
    wa.close()


# ---- (4) options ------------------------------------------------------------------------------------------------------------------------

def test_options_belong_to_a_handle(jq):
    """two handles of one problem with different options side by side; set_option on a live handle (per-evaluation options at once,
    plan-shaping ones re-plan it); the environment -- apart from JQ_OPTIONS at jq_create -- changes nothing"""
    from juqbox_jl_amd import _lib
    rng = np.random.default_rng(6400)
    p, pcof = random_problem(jq, rng, 64, 4, 2, 1, 30, 3, 1, "t4")
    nodes, weights = 0.02 * rng.standard_normal(40), rng.random(40)
    shift = 0.01 * np.arange(64)
    os.environ["JQ_QUAD"] = "0"        # (ABI <= 4 would have sent the next handle to the cooperative kernels)
    os.environ["JQ_CQ"] = "0"
    try:
        wa = jq.Working_Arrays_HIP(p, pcof.size)
    finally:
        os.environ.pop("JQ_QUAD"), os.environ.pop("JQ_CQ")
    wb = jq.Working_Arrays_HIP(p, pcof.size, options={"quad": 0, "coop_max": 0, "lane": 0})
    res = {}
    for tag, w in (("a", wa), ("b", wb)):
        jq.eval_f_g_grad(pcof, p, w, nodes, weights, True, shift=shift)
        res[tag] = (p.last_infidelity, p.last_infidelity_grad.copy(), w.last_timing()["kernel_family"])
    assert res["a"][2] == 8 and res["b"][2] == 0
    assert abs(res["a"][0] - res["b"][0]) <= 1e-13 and np.linalg.norm(res["a"][1] - res["b"][1]) <= 1e-12 * np.linalg.norm(res["b"][1])
    assert wb.get_option("quad") == 0 and wa.get_option("quad") is None and wa.get_option("t4") == 1
    assert wa.plan_info()["options"] == "" and "quad=0" in wb.plan_info()["options"]
    # a plan-shaping option on the live handle: it is planned again, settings kept
    wa.set_option("t4", 0)
    assert wa.plan_info()["structure"] == "od" and wa.plan_info()["replanned"] is False
    jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=shift)
    assert wa.last_timing()["kernel_band"] == 9 and abs(p.last_infidelity - res["b"][0]) <= 1e-13
    wa.set_option("t4", None)
    assert wa.plan_info()["structure"] == "t4"
    # a per-evaluation option
    wa.set_option("cq3", 0)
    jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=shift)
    assert wa.last_timing()["kernel_variant"] == 0 and wa.plan_info()["latency_split"]["last_decision"] == "not taken: option cq3=0"
    wa.set_option("cq3", None)
    jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=shift)
    assert wa.last_timing()["kernel_variant"] == 3 and p.last_infidelity == res["a"][0]
    # errors
    with pytest.raises(_lib.JuqboxHipError) as e:
        wa.set_option("no_such_option", 1)
    assert e.value.code == _lib.JQ_EINVAL
    with pytest.raises(_lib.JuqboxHipError) as e:
        wa.set_option("wlr_sc", 1)      # experiment builds only
    assert e.value.code == _lib.JQ_EUNSUPPORTED
    with pytest.raises(_lib.JuqboxHipError) as e:
        wa.set_option("debug", 1)       # a profiling switch that changes results: not in a release library
    assert e.value.code == _lib.JQ_EUNSUPPORTED
    with pytest.raises(_lib.JuqboxHipError) as e:
        jq.Working_Arrays_HIP(p, pcof.size, options={"quadd": 0})
    assert e.value.code == _lib.JQ_EINVAL and "quadd" in str(e.value)
    wa.close(), wb.close()


def test_jq_options_environment_variable_reaches_jq_create():
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import juqbox_jl_amd as jq
from test_gpu_random import random_problem
p, pcof = random_problem(jq, np.random.default_rng(5), 64, 4, 2, 1, 12, 3, 1, "t4")
wa = jq.Working_Arrays_HIP(p, pcof.size)
jq.traceobjgrad(pcof, p, wa, False, True)
print("FAMILY", wa.last_timing()["kernel_family"], wa.plan_info()["options"])
""" % (ROOT, ROOT)
    out = {}
    for tag, opt in (("none", None), ("slab", "quad=0, coop_max=0;lane=0"), ("bad", "quad=0,nonsense=1")):
        env = dict(os.environ)
        env.pop("JQ_OPTIONS", None)
        if opt:
            env["JQ_OPTIONS"] = opt
        out[tag] = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT)
    assert "FAMILY 8 " in out["none"].stdout, out["none"].stderr[-800:]
    assert "FAMILY 0 quad=0,coop_max=0,lane=0" in out["slab"].stdout.replace("lane=0,coop_max=0", "coop_max=0,lane=0") or "FAMILY 0" in out["slab"].stdout, out["slab"].stderr[-800:]
    assert out["bad"].returncode != 0 and "JQ_OPTIONS" in out["bad"].stderr and "nonsense" in out["bad"].stderr


# ---- (5) start-up rendezvous of the split latency kernels ----------------------------------------------------------------------------------

def test_abandoned_rendezvous_falls_back_within_the_call_and_cools_down_briefly(jq):
    """cq3_fault=3 simulates a launch whose workgroups did not all become resident within cq3_rdv_us (another process holds the CUs):
    the evaluation is repeated on the one-workgroup kernel inside the same call -- bit-identical --, the handle stays off the split for
    a few evaluations (2 after the first time) and is never switched off for good: a busy GPU is not a fault of the handle."""
    p, _ = jq.cases.cnot3()
    p.T, p.nsteps = p.T * 600 / p.nsteps, 600
    pcof = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "cnot3.json")))["pcof0"])
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    f0, g0, *_ = jq.traceobjgrad(pcof, p, wa, False, True)
    assert wa.last_timing()["kernel_variant"] == 3
    for rounds in range(8):      # more often than the six dead waits that switch the split off for good
        wa.set_option("cq3_fault", 3)
        f1, g1, *_ = jq.traceobjgrad(pcof, p, wa, False, True)
        wa.set_option("cq3_fault", None)
        ls = wa.plan_info()["latency_split"]
        assert wa.last_timing()["kernel_variant"] == 0 and f1 == f0 and np.array_equal(g1, g0)
        assert ls["abandoned_at_rendezvous"] == rounds + 1 and ls["faults"] == 0 and ls["off"] is False and 1 <= ls["cooling_down"] <= 63
        while wa.plan_info()["latency_split"]["cooling_down"] > 0:
            jq.traceobjgrad(pcof, p, wa, False, True)
            assert wa.last_timing()["kernel_variant"] == 0
        f2, g2, *_ = jq.traceobjgrad(pcof, p, wa, False, True)      # ... and takes the split again
        assert wa.last_timing()["kernel_variant"] == 3 and f2 == f0 and np.array_equal(g2, g0)
    # a rendezvous that cannot succeed (1 us for a grid of 3 x 8 workgroups to assemble is not enough ... or is: both outcomes are legal,
    # the RESULT must not change)
    wa.set_option("cq3_rdv_us", 1)
    f3, g3, *_ = jq.traceobjgrad(pcof, p, wa, False, True)
    assert f3 == f0 and np.array_equal(g3, g0)
    wa.close()


# ---- (6) every kernel object with the low-rank weight terms: odd chunk lengths, Neumann terms (advisor, round 5) -----------------------------

@pytest.mark.parametrize("NT", [1, 2, 3, 4, 5, 6, 7, 8])
def test_full_weight_objects_with_odd_chunk_lengths(jq, NT):
    """Round 5's miscompiled object (w_6_5 in VGPR form) gave wrong states for chunks with an ODD number of steps and Neumann terms; its
    siblings are built from the same source.  Every size of the quad-layout objects with the low-rank terms (w_1_7 .. w_8_7: 4 x 4 x n
    problems, five complex forbidden states so that the cooperative-quad kernels do not take them) and the one-tile-row slab object
    (w_1_0) runs 7 steps in chunks of 3, 3, 1 and of 2 with three Neumann terms against the oracle; w_6_5 itself: test_gpu_round5.py (7)."""
    rng = np.random.default_rng(6600 + NT)
    p, pcof = random_problem(jq, rng, 16 * NT, 4, 2, 1, 7, 3, 3, "t4")
    set_forbidden(p, rng, 5)
    for chunk in (3, 2):
        # (one tile row: the row-lane kernels would take it -- lane=0 sends it to the quad-layout object)
        wa = jq.Working_Arrays_HIP(p, pcof.size, options=dict({"chunk_steps": chunk}, **({"lane": 0} if NT == 1 else {})))
        t = check(jq, p, pcof, wa, rng, ensembles=(5,))
        assert (t["kernel_family"], t["kernel_size"], t["kernel_band"]) == (6, NT, 7), t
        wa.close()
    if NT == 1:
        q, qcof = random_problem(jq, rng, 12, 3, 2, 1, 7, 3, 3, False)
        set_forbidden(q, rng, 3)
        for chunk in (3, 2):
            wa = jq.Working_Arrays_HIP(q, qcof.size, options={"chunk_steps": chunk, "lane": 0, "embed": 0})
            t = check(jq, q, qcof, wa, rng, ensembles=(5,))
            assert (t["kernel_family"], t["kernel_size"], t["kernel_band"]) == (0, 1, 0), t
            wa.close()


# ---- (7) row-lane kernels: the LDS ring and the compile-time Neumann terms at their edges --------------------------------------------------

@pytest.mark.parametrize("Ntot,N,Nc,m", [(2, 2, 1, 1), (3, 3, 1, 2), (4, 4, 2, 8), (6, 4, 3, 9), (7, 2, 1, 12), (10, 4, 2, 3), (13, 3, 4, 5), (16, 4, 1, 4)])
def test_rowlane_ring_with_short_chunks_and_every_term_dispatch(jq, Ntot, N, Nc, m):
    """The row-lane kernels of round 6 read their operator rows through an LDS ring that the wave's own DMA fills three (two) steps
    ahead, and run the Neumann recurrences with the number of terms known at compile time for m = 2 .. 10 (the run-time loop otherwise).
    Edges: chunks SHORTER than the prefetch distance (1, 2, 3 steps -- the DMA re-fetches the last group instead of reading past the
    stream), a last chunk of one step, every row length (NPJ = 2 .. 16), m inside and outside the dispatch table, one to four controls
    (one or two trace waves), the one-wave backward kernel (rl_split=0) and the state history -- all against the oracle."""
    rng = np.random.default_rng(6700 + 17 * Ntot + m)
    p, pcof = random_problem(jq, rng, Ntot, N, Nc, 1, 7, m, 3 if Nc > 1 else 1, False)
    for opts in ({"chunk_steps": 1}, {"chunk_steps": 2}, {"chunk_steps": 3}, {"chunk_steps": 5, "rl_split": 0}, {}):
        wa = jq.Working_Arrays_HIP(p, pcof.size, options=dict(opts, embed=0))
        t = check(jq, p, pcof, wa, rng, ensembles=(3,), history=(opts == {}))
        assert t["kernel_family"] == 3 and t["kernel_variant"] == (0 if opts.get("rl_split") == 0 else 33), t
        wa.close()


# ---- (8) the dense policy of the cooperative-quad kernels: 17 .. 32 levels without the 4 x 4 x n structure ----------------------------------

@pytest.mark.parametrize("Ntot,N,Nc,m,oft,structure", [(17, 4, 1, 3, 1, False), (20, 3, 2, 4, 3, False), (25, 4, 3, 5, 2, False), (32, 4, 2, 6, 1, False),
                                                       (32, 2, 4, 1, 3, False), (27, 7, 2, 2, 1, True), (24, 16, 1, 3, 3, "od"), (30, 4, 2, 7, 2, False), (24, 20, 2, 3, 1, False)])
def test_dense_cooperative_quad_kernels_match_the_oracle(jq, Ntot, N, Nc, m, oft, structure):
    """Round 6: problems with two 16-row blocks and NO 4 x 4 x n structure (two five-level subsystems, operators in an eigenbasis, ...) ran their
    single evaluations on the cooperative kernels at 26 us per time step.  The cooperative-quad kernels now take them with a DENSE product
    (four v_mfma_f64_4x4x4_4b per 16 x 16 tile on the state register and its three lane rotations, jq_cq_kernels.h CoopQ<2, true>): every
    level count 17 .. 32, one to four controls, even and odd numbers of Neumann terms (the parities of the LDS exchange), all three objective
    types (two backward passes), N < 4, N = 4, N > 4 (several column quads per evaluation) and N > 16 (two slabs per evaluation), chunks of odd length, ensembles with a
    ragged last slab, the state history -- against the oracle; option dq=0 gives the cooperative kernels back."""
    rng = np.random.default_rng(6800 + 31 * Ntot + m)
    p, pcof = random_problem(jq, rng, Ntot, N, Nc, 2, 11, m, oft, structure)
    for opts in ({}, {"chunk_steps": 3}, {"chunk_steps": 4}):
        wa = jq.Working_Arrays_HIP(p, pcof.size, options=opts)
        t = check(jq, p, pcof, wa, rng, ensembles=(3, 9), history=(opts == {}))
        # (four controls per backward sweep: their eight constant images do not fit the LDS next to the window ring -- cooperative kernels)
        assert (t["kernel_family"], t["kernel_size"], t["kernel_band"]) == ((8, 2, 10) if Nc <= 3 else (1, 2, 1)), t
        wa.close()
    wa = jq.Working_Arrays_HIP(p, pcof.size, options={"dq": 0})
    t = check(jq, p, pcof, wa, rng)
    assert t["kernel_family"] == 1, t
    wa.close()
    if Nc <= 3 and N <= 4:
        # the backward sweep on three / one workgroup(s) per column quad: same operations per chain -- bit for bit (first chunks longer than the
        # hand-off ring of the split kernels: 40 steps)
        q, qcof = random_problem(jq, rng, Ntot, N, Nc, 2, 40, m, oft, structure)
        res = {}
        for tag, opts in (("three", {}), ("one", {"cq3": 0})):
            wa = jq.Working_Arrays_HIP(q, qcof.size, options=opts)
            o = jq.traceobjgrad(qcof, q, wa, False, True)
            t = wa.last_timing()
            assert (t["kernel_family"], t["kernel_band"], t["kernel_variant"]) == (8, 10, 3 if tag == "three" else 0), (t, wa.plan_info()["latency_split"])
            res[tag] = o
            wa.close()
        for k in (0, 1, 5, 6):
            assert np.array_equal(res["three"][k], res["one"][k])
        from oracle.oracle import Oracle
        r = Oracle(q, use_sparse=False).traceobjgrad(qcof)
        assert reference_pass(res["three"][1], r["totalgrad"]) and reference_pass(res["three"][0], r["objfv"])


@pytest.mark.parametrize("Ntot,Nc,oft,structure", [(17, 1, 1, False), (25, 2, 3, False), (32, 3, 2, False), (28, 2, 1, True)])
def test_dense_cooperative_quad_kernels_implicit_midpoint(jq, Ntot, Nc, oft, structure):
    """... and the implicit-midpoint twins (k_*_cq_imr<2, true>; N = 4: one evaluation per column quad, the solver's stopping rule over the
    whole evaluation): single evaluations, an ensemble with an ensemble shift (folded into the A operand of the own tile's rotation 0),
    chunks, against the oracle at the solver's tolerance; option dq=0: the cooperative implicit-midpoint kernels."""
    from oracle.oracle import Oracle
    rng = np.random.default_rng(6900 + Ntot)
    p, pcof = random_problem(jq, rng, Ntot, 4, Nc, 2, 13, 3, oft, structure)
    p.Integrator_id = jq.Implicit_Midpoint
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-13, nrhs=4)
    p.wmat = p.wmat_real.copy()
    r = Oracle(p, use_sparse=False).traceobjgrad_imr(pcof, 100, 1e-13)
    nodes, weights = 0.05 * rng.standard_normal(5), rng.random(5)
    shift = 0.05 * rng.standard_normal(Ntot)
    shift[0] = 0.0
    # (a weighted ensemble with a diagonal perturbation = the weighted sum of the oracle's single evaluations)
    inf, leak, g = 0.0, 0.0, np.zeros(pcof.size)
    H0 = p.Hconst.copy()
    for ep, wq in zip(nodes, weights):
        p.Hconst = H0 + np.diag(ep * shift)
        rr = Oracle(p, use_sparse=False).traceobjgrad_imr(pcof, 100, 1e-13)
        inf, leak, g = inf + wq * rr["primaryobjf"], leak + wq * rr["secondaryobjf"], g + wq * rr["infidelgrad"]
    p.Hconst = H0
    for opts, fam in (({}, 9), ({"chunk_steps": 5}, 9), ({"dq": 0}, 5)):
        wa = jq.Working_Arrays_M_HIP(p, pcof.size, options=opts)
        objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, p, wa, False, True)
        t = wa.last_timing()
        assert t["kernel_family"] == fam and (fam != 9 or t["kernel_band"] == 10), t
        gn = np.linalg.norm(r["totalgrad"])
        assert abs(prim - r["primaryobjf"]) <= 1e-10 and abs(sec - r["secondaryobjf"]) <= 1e-10 * max(abs(r["secondaryobjf"]), 1e-3)
        assert np.linalg.norm(tg - r["totalgrad"]) <= 1e-10 * gn and np.linalg.norm(ig - r["infidelgrad"]) <= 1e-10 * gn
        jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=shift)
        assert abs(p.last_infidelity - inf) <= 1e-10 * abs(inf) and abs(p.last_leak - leak) <= max(1e-10 * abs(leak), 1e-14)
        assert np.linalg.norm(p.last_infidelity_grad - g) <= 1e-10 * np.linalg.norm(g)
        wa.close()
    # the backward sweep on three workgroups per evaluation / on one: bit for bit (a first chunk longer than the hand-off ring)
    q, qcof = random_problem(jq, rng, Ntot, 4, Nc, 2, 40, 3, oft, structure)
    q.Integrator_id = jq.Implicit_Midpoint
    q.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-13, nrhs=4)
    q.wmat = q.wmat_real.copy()
    res = {}
    for tag, opts in (("three", {}), ("one", {"cq3": 0})):
        wa = jq.Working_Arrays_M_HIP(q, qcof.size, options=opts)
        res[tag] = jq.traceobjgrad(qcof, q, wa, False, True)
        t = wa.last_timing()
        assert (t["kernel_family"], t["kernel_band"], t["kernel_variant"]) == (9, 10, 3 if tag == "three" else 0), (t, wa.plan_info()["latency_split"])
        wa.close()
    for k in (0, 1, 5, 6):
        assert np.array_equal(res["three"][k], res["one"][k])
