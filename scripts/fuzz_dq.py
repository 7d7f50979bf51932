"""Randomised stress of the DENSE cooperative-quad policy (17 .. 32 levels without the 4 x 4 x n structure; jq_cq_kernels.h CoopQ<2, true>)
against the CPU oracle: level count, columns, controls, Neumann terms (even / odd: the parities of the LDS exchange), objective type,
chunking, ensembles up to three rounds of workgroups, both integrators (implicit midpoint: N = 4), three / one workgroup(s) per column
quad in the backward sweep.  usage: fuzz_dq.py [n_cases] [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from oracle.oracle import Oracle
from test_gpu_random import random_problem
sys.path.insert(0, "scripts")
from fuzz_gpu import ref_err, RTOL

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4545)
worst, t0, fams = 0.0, time.time(), {}
for case in range(n_cases):
    imr = rng.random() < 0.3
    Ntot = int(rng.integers(17, 33))
    N = 4 if imr else int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8]))
    Nc, m, oft = int(rng.integers(1, 4)), int(rng.integers(1, 9)), int(rng.integers(1, 4))
    nsteps = int(rng.integers(3, 70))
    structure = rng.choice([False, False, True, "od"])
    structure = structure if structure in ("od",) else bool(structure == "True")
    p, pcof = random_problem(jq, rng, Ntot, N, Nc, int(rng.integers(1, 3)), nsteps, m, oft, structure)
    opts = {}
    if rng.random() < 0.5:
        opts["chunk_steps"] = int(rng.integers(1, nsteps + 1))
    if rng.random() < 0.3:
        opts["cq3"] = 0
    nq = int(rng.choice([1, 1, 3, 9, 70, 300]))
    nodes, weights = 0.05 * rng.standard_normal(nq), rng.random(nq)
    shift = 0.05 * rng.standard_normal(Ntot)
    shift[0] = 0.0
    if imr:
        p.Integrator_id = jq.Implicit_Midpoint
        p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-13, nrhs=N)
        p.wmat = p.wmat_real.copy()
        wa = jq.Working_Arrays_M_HIP(p, pcof.size, options=opts)
        r = Oracle(p, use_sparse=False).traceobjgrad_imr(pcof, 100, 1e-13)
    else:
        wa = jq.Working_Arrays_HIP(p, pcof.size, options=opts)
        r = Oracle(p, use_sparse=False).traceobjgrad(pcof)
    o = jq.traceobjgrad(pcof, p, wa, False, True)
    t = wa.last_timing()
    err = max(ref_err(o[2], r["primaryobjf"]), ref_err(o[3], r["secondaryobjf"]), ref_err(o[1], r["totalgrad"]), ref_err(o[5], r["infidelgrad"]))
    if oft != 1:
        err = max(err, ref_err(o[6], r["leakgrad"]))
    if not imr and nq > 1:
        ref = Oracle(p, use_sparse=False).eval_f_g_grad(pcof, nodes, weights, shift)
        jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=shift)
        err = max(err, ref_err(p.last_infidelity, ref["last_infidelity"]), ref_err(p.last_leak, ref["last_leak"]), ref_err(p.last_infidelity_grad, ref["last_infidelity_grad"]))
        t = wa.last_timing()
    wa.close()
    key = (t["kernel_family"], t["kernel_band"], t["kernel_variant"])
    fams[key] = fams.get(key, 0) + 1
    tol = 1e-9 if imr else RTOL      # (implicit midpoint: the solver's tolerance 1e-13 per step bounds the agreement)
    flag = "" if err < tol else "   <-- MISMATCH"
    print("%4d Ntot=%2d N=%d Nc=%d m=%d oft=%d steps=%2d %-5s %s nq=%3d %-28s fam=%s err=%.1e%s" % (case, Ntot, N, Nc, m, oft, nsteps, structure, "IMR" if imr else "SV ", nq, str(opts), key, err, flag), flush=True)
    worst = max(worst, err)
print("worst relative error %.2e over %d cases in %.0f s; (family, band, variant): %s" % (worst, n_cases, time.time() - t0, sorted(fams.items())))
