"""Why do three rounds of the implicit-midpoint quad kernels take more than 3 x one round?  cnot3, full length."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import case_inputs
import juqbox_jl_amd as jq

params, info, pcof, _ = case_inputs("cnot3")
ns_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if ns_steps:
    params.T = params.T * ns_steps / params.nsteps
    params.nsteps = ns_steps
params.Integrator_id = jq.Implicit_Midpoint
params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
wa = jq.Working_Arrays_M_HIP(params, pcof.size)
for ns, same in ((1024, False), (1024, True), (2048, False), (2048, True), (3072, False), (3072, True), (1280, False)):
    nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
    if same:
        nodes = np.full(ns, nodes[ns // 3])
    for adj in (False, True):
        jq.eval_f_g_grad(pcof, params, wa, nodes, weights, adj, shift=shift)
        t0 = time.perf_counter()
        jq.eval_f_g_grad(pcof, params, wa, nodes, weights, adj, shift=shift)
        dt = time.perf_counter() - t0
        t = wa.last_timing()
        print("%5d samples %s adjoint=%d  %.3f s  fwd %.1f ms (%d launches) bwd %.1f ms (%d launches)" %
              (ns, "same " if same else "mixed", adj, dt, t["ms_forward"], t["n_forward_launches"], t["ms_backward"], t["n_backward_launches"]), flush=True)
