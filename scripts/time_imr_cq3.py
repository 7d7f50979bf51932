"""Implicit midpoint, latency path: k_backward_cq_imr3 (three workgroups per evaluation) next to the two-set kernel on one CU (JQ_CQ3=0)
and the one-set kernel (JQ_CQ3=0 JQ_IMR_CQ2=0): times and bit-wise comparison.  python scripts/time_imr_cq3.py [nsteps] [sizes]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juqbox_jl_amd as jq
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 9, 80]
params, info = jq.cases.cnot3()
if nsteps:
    params.T, params.nsteps = params.T * nsteps / params.nsteps, nsteps
params.Integrator_id = jq.Implicit_Midpoint
params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
pcof = np.array(json.load(open(os.path.join(ROOT, "tests/golden/cnot3.json")))["pcof0"])
res = {}
for tag, env in (("three", {}), ("two", {"JQ_CQ3": "0"}), ("one", {"JQ_CQ3": "0", "JQ_IMR_CQ2": "0"})):
    os.environ.update(env)
    wa = jq.Working_Arrays_M_HIP(params, pcof.size)
    for ns in sizes:
        nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
        best = None
        for rep in range(2):
            jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
            t = wa.last_timing()
            best = t if best is None or t["ms_total"] < best["ms_total"] else best
        res[tag, ns] = (params.last_infidelity, params.last_leak, params.last_infidelity_grad.copy())
        print("%-6s %4d samples: family %d variant %d  %.1f ms (fwd %.1f bwd %.1f)" %
              (tag, ns, best["kernel_family"], best["reserved"], best["ms_total"], best["ms_forward"], best["ms_backward"]), flush=True)
    wa.close()
    for k in env:
        os.environ.pop(k, None)
for ns in sizes:
    a = res["three", ns]
    for other in ("two", "one"):
        b = res[other, ns]
        print("%4d samples, three vs %s: bit-identical %s (gradient diff %.1e)" % (ns, other, a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]), np.linalg.norm(a[2] - b[2])))
