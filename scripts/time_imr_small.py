"""Single-evaluation latency of the implicit-midpoint path for the small reference cases (row-lane kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import case_inputs
import juqbox_jl_amd as jq
for case in ("swap02", "flux", "cnot1", "cnot2", "cnot2-leakieq"):
    params, info, pcof, _ = case_inputs(case)
    params.Integrator_id = jq.Implicit_Midpoint
    params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
    wa = jq.Working_Arrays_M_HIP(params, pcof.size)
    jq.traceobjgrad(pcof, params, wa, False, True)
    t0 = time.perf_counter(); r = jq.traceobjgrad(pcof, params, wa, False, True); dt = time.perf_counter() - t0
    t = wa.last_timing()
    print("%-14s Ntot=%2d  %.1f ms (fwd %.1f bwd %.1f) family %d objf %.15e" % (case, params.Ntot, dt * 1e3, t["ms_forward"], t["ms_backward"], t["kernel_family"], r[0]), flush=True)
    wa.close()
