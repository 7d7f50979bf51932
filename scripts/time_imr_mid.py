import sys, time, copy
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from oracle.oracle import Oracle
from test_gpu_random import random_problem
for Ntot in (25, 32, 48):
    rng = np.random.default_rng(2525 + Ntot)
    p, pcof = random_problem(jq, rng, Ntot, 4, 2, 2, 2000, 4, 1, False)
    p.Integrator_id = jq.Implicit_Midpoint
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=4)
    p.wmat = p.wmat_real.copy()
    wa = jq.Working_Arrays_M_HIP(p, pcof.size)
    for _ in range(2):
        jq.traceobjgrad(pcof, p, wa)
    t = wa.last_timing()
    t0 = time.perf_counter(); Oracle(p, use_sparse=False).traceobjgrad_imr(pcof, 100, 1e-12); tc = time.perf_counter() - t0
    print("IMR Ntot %d: family %d <%d,%d> fwd %.2f bwd %.2f total %.2f ms = %.2f us/step | CPU %.1f ms (%.1f x)" % (Ntot, t["kernel_family"], t["kernel_size"], t["kernel_band"], t["ms_forward"], t["ms_backward"], t["ms_total"], t["ms_total"] * 1e3 / 2000, tc * 1e3, tc * 1e3 / t["ms_total"]), flush=True)
    wa.close()
