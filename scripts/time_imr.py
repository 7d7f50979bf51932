"""Single-evaluation latency of the implicit-midpoint path at cnot3 on the quad-layout kernels (default) and on the cooperative
kernels (JQ_QUAD=0), next to the Stormer-Verlet path.  python scripts/time_imr.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from conftest import case_inputs  # noqa: E402
import juqbox_jl_amd as jq  # noqa: E402


def run(tag, imr, env):
    params, info, pcof, _ = case_inputs("cnot3")
    if imr:
        params.Integrator_id = jq.Implicit_Midpoint
        params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
    os.environ.update(env)
    try:
        wa = (jq.Working_Arrays_M_HIP if imr else jq.Working_Arrays_HIP)(params, pcof.size)
        jq.traceobjgrad(pcof, params, wa, False, True)
        t0 = time.perf_counter()
        r = jq.traceobjgrad(pcof, params, wa, False, True)
        dt = time.perf_counter() - t0
    finally:
        for k in env:
            os.environ.pop(k, None)
    t = wa.last_timing()
    print("%-28s %.3f s  (forward %.1f ms, backward %.1f ms, family %d)  objf %.15e  |grad| %.15e" %
          (tag, dt, t["ms_forward"], t["ms_backward"], t["kernel_family"], r[0], float(np.linalg.norm(r[1]))), flush=True)
    wa.close()


run("Stormer-Verlet", False, {})
run("implicit midpoint, cq", True, {})
run("implicit midpoint, cq one set", True, {"JQ_IMR_CQ2": "0"})
run("implicit midpoint, quad", True, {"JQ_IMR_CQ": "0"})
run("implicit midpoint, coop", True, {"JQ_QUAD": "0"})
