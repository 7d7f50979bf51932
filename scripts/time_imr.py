"""Development aid: timings of the implicit-midpoint path (GPU) next to the CPU oracle."""
import json, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from oracle.oracle import Oracle
cases = sys.argv[1:] or ["swap02", "flux", "cnot1", "cnot2"]     # (cnot3: the CPU oracle takes ~18 s)
for case in cases:
    p, info = jq.cases.BUILDERS[case]()
    g = json.load(open("tests/golden/%s.json" % info["golden"])) if info.get("golden") else None
    pcof = np.array(g["pcof0"]) if g and "pcof0" in g else info["pcof0"]
    p.Integrator_id = jq.Implicit_Midpoint
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=p.N)
    wa = jq.Working_Arrays_M_HIP(p, pcof.size)
    jq.traceobjgrad(pcof, p, wa)
    t = wa.last_timing()
    t0 = time.perf_counter(); Oracle(p).traceobjgrad_imr(pcof, 100, 1e-12); tc = time.perf_counter() - t0
    x, w = np.polynomial.legendre.leggauss(512)
    shift = p.shift_weights_reference() if p.Ntot <= 4 else 0.01 * np.arange(p.Ntot)
    jq.eval_f_g_grad(pcof, p, wa, x * 0.05, w * 0.5, True, shift=shift)
    t2 = wa.last_timing()
    print("%-8s Ntot=%2d nsteps=%5d  GPU single %.1f ms (fwd %.1f bwd %.1f) | CPU oracle %.0f ms | GPU 512 samples %.1f ms" % (
        case, p.Ntot, p.nsteps, t["ms_total"], t["ms_forward"], t["ms_backward"], tc * 1e3, t2["ms_total"]), flush=True)
    wa.close()
