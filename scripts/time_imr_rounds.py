"""Cost of one publication round of the cooperative-quad implicit-midpoint kernels: cnot3 with the iteration count pinned by
max_iter (tol = 1e-300: never converged) -- time per evaluation is linear in it.  python scripts/time_imr_rounds.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from conftest import case_inputs  # noqa: E402
import juqbox_jl_amd as jq  # noqa: E402

res = []
for max_iter, tol in ((2, 1e-300), (4, 1e-300), (8, 1e-300), (12, 1e-300), (100, 1e-12)):
    params, info, pcof, _ = case_inputs("cnot3")
    params.Integrator_id = jq.Implicit_Midpoint
    params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=max_iter, tol=tol, nrhs=params.N)
    wa = jq.Working_Arrays_M_HIP(params, pcof.size)
    jq.traceobjgrad(pcof, params, wa, False, True)
    jq.traceobjgrad(pcof, params, wa, False, True)
    t = wa.last_timing()
    res.append((max_iter, tol, t["ms_forward"], t["ms_backward"]))
    print("max_iter %3d tol %g: forward %.1f ms, backward %.1f ms (family %d)" % (max_iter, tol, t["ms_forward"], t["ms_backward"], t["kernel_family"]), flush=True)
    wa.close()
ns = 32386
f = np.polyfit([r[0] for r in res[:4]], [r[2] for r in res[:4]], 1)
b = np.polyfit([r[0] for r in res[:4]], [r[3] for r in res[:4]], 1)
print("forward: %.0f clk per round, %.0f clk per step besides  (2.4 GHz)" % (f[0] * 1e-3 / ns * 2.4e9, f[1] * 1e-3 / ns * 2.4e9))
print("backward: %.0f clk per round pair, %.0f clk per step besides" % (b[0] * 1e-3 / ns * 2.4e9, b[1] * 1e-3 / ns * 2.4e9))
print("golden run: forward = %.2f iterations, backward = %.2f" % ((res[4][2] - f[1]) / f[0], (res[4][3] - b[1]) / b[0]))
