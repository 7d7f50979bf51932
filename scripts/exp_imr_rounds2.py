import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import case_inputs
import juqbox_jl_amd as jq
params, info, pcof, _ = case_inputs("cnot3")
ns_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
params.T = params.T * ns_steps / params.nsteps
params.nsteps = ns_steps
params.Integrator_id = jq.Implicit_Midpoint
params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
wa = jq.Working_Arrays_M_HIP(params, pcof.size)
for ns in (1024, 2048, 2304, 2560, 2816, 3072, 3328, 4096, 5120, 6144):
    nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, False, shift=shift)
    t0 = time.perf_counter()
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, False, shift=shift)
    dt = time.perf_counter() - t0
    t = wa.last_timing()
    print("%5d samples (%4d workgroups)  fwd %.1f ms (%d launches)  = %.2f rounds" % (ns, ns // 4, t["ms_forward"], t["n_forward_launches"], 0), flush=True)
