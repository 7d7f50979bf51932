"""cnot2 with the Jacobi solver of the Stormer-Verlet path (golden cnot2-jacobi) vs the Neumann solver: kernel family and time."""
import json, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from conftest import case_inputs
for case in ("cnot2", "cnot2-jacobi"):
    params, info, pcof, g = case_inputs(case)
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    for ns in (1, 512):
        x, w = np.polynomial.legendre.leggauss(ns)
        for rep in range(2):
            jq.eval_f_g_grad(pcof, params, wa, x * 0.05 if ns > 1 else np.zeros(1), w * 0.5 if ns > 1 else np.ones(1), True, shift=0.01 * np.arange(params.Ntot))
        t = wa.last_timing()
        print("%-13s solver %d  %4d samples: family %d <%d,%d>  %.1f ms" % (case, params.linear_solver.solver_id, ns, t["kernel_family"], t["kernel_size"], t["kernel_band"], t["ms_total"]), flush=True)
    wa.close()
