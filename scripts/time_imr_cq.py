"""One cnot3 evaluation with the implicit-midpoint integrator on the cooperative-quad kernels (for rocprofv3 --kernel-trace --stats).
python scripts/time_imr_cq.py [nsamples]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from conftest import case_inputs  # noqa: E402
import juqbox_jl_amd as jq  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 0
params, info, pcof, _ = case_inputs("cnot3")
params.Integrator_id = jq.Implicit_Midpoint
params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
wa = jq.Working_Arrays_M_HIP(params, pcof.size)
for rep in range(2):
    t0 = time.perf_counter()
    if ns:
        nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
        r = jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
    else:
        r = jq.traceobjgrad(pcof, params, wa, False, True)
    dt = time.perf_counter() - t0
    t = wa.last_timing()
    print("%d samples: %.3f s (forward %.1f ms, backward %.1f ms, family %d)" % (max(ns, 1), dt, t["ms_forward"], t["ms_backward"], t["kernel_family"]), flush=True)
wa.close()
