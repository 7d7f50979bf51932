"""Implicit-midpoint ENSEMBLE throughput at cnot3 (the reference examples' default integrator): evals/s for several ensemble
sizes on the kernels run_eval picks (cooperative-quad up to 512 column quads, quad layout beyond), next to Stormer-Verlet.
python scripts/time_imr_ensemble.py [nsteps]   (default: the full 32 386 steps)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from conftest import case_inputs  # noqa: E402
import juqbox_jl_amd as jq  # noqa: E402

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 512, 1024, 2048, 3072]


def run(tag, imr, ns):
    params, info, pcof, _ = case_inputs("cnot3")
    if nsteps:
        params.T = params.T * nsteps / params.nsteps
        params.nsteps = nsteps
    if imr:
        params.Integrator_id = jq.Implicit_Midpoint
        params.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=params.N)
    nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
    wa = (jq.Working_Arrays_M_HIP if imr else jq.Working_Arrays_HIP)(params, pcof.size)
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
    t0 = time.perf_counter()
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
    dt = time.perf_counter() - t0
    t = wa.last_timing()
    print("%-20s %5d samples  %.3f s  %8.1f evals/s (x %d/32386 steps)  fwd %.1f ms bwd %.1f ms  family %d  infid %.15e" %
          (tag, ns, dt, ns / dt, params.nsteps, t["ms_forward"], t["ms_backward"], t["kernel_family"], params.last_infidelity), flush=True)
    wa.close()


for ns in sizes:
    run("implicit midpoint", True, ns)
for ns in sizes[-1:]:
    run("Stormer-Verlet", False, ns)
