"""Soak of the three-workgroup latency kernels: many evaluations of the same problems, every result compared bit for bit with the
first one (a race between the workgroups of a quad would show up as a rare mismatch).  python scripts/soak_cq3.py [rounds] [nsteps]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juqbox_jl_amd as jq
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
pcof = np.array(json.load(open(os.path.join(ROOT, "tests/golden/cnot3.json")))["pcof0"])
bad = 0
t0 = time.time()
for imr in (False, True):
    params, info = jq.cases.cnot3()
    params.T, params.nsteps = params.T * nsteps / params.nsteps, nsteps
    if imr:
        params.Integrator_id = jq.Implicit_Midpoint
        params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
    wa = (jq.Working_Arrays_M_HIP if imr else jq.Working_Arrays_HIP)(params, pcof.size)
    first = {}
    for r in range(rounds):
        for ns in (1, 9, 80):
            nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
            jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
            assert wa.last_timing()["reserved"] == 3
            cur = (params.last_infidelity, params.last_leak, params.last_infidelity_grad.tobytes())
            if ns not in first:
                first[ns] = cur
            elif cur != first[ns]:
                bad += 1
                print("MISMATCH imr=%s round %d ns %d" % (imr, r, ns), flush=True)
    wa.close()
print("%d rounds x 3 ensemble sizes x 2 integrators x %d steps: %d mismatches in %.0f s" % (rounds, nsteps, bad, time.time() - t0))
