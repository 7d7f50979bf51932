"""Implicit-midpoint ensembles at cnot3: seconds per ensemble evaluation vs. the number of samples.  python scripts/time_imr_sizes.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from conftest import case_inputs  # noqa: E402
import juqbox_jl_amd as jq  # noqa: E402

params, info, pcof, _ = case_inputs("cnot3")
params.Integrator_id = jq.Implicit_Midpoint
params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
wa = jq.Working_Arrays_M_HIP(params, pcof.size)
for ns in [int(a) for a in sys.argv[1:]] or [1, 64, 256, 512, 513, 1024, 3072]:
    nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
    t0 = time.perf_counter()
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
    dt = time.perf_counter() - t0
    t = wa.last_timing()
    print("%5d samples: %.3f s = %7.1f evals/s  (family %d, forward %.0f ms, backward %.0f ms)" %
          (ns, dt, ns / dt, t["kernel_family"], t["ms_forward"], t["ms_backward"]), flush=True)
wa.close()
