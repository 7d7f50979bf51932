"""Development aid: forward / backward time of single implicit-midpoint evaluations of the small reference cases (row-lane kernels)."""
import copy, json, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
for case in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["swap02", "flux", "cnot1", "cnot2"]):
    p, info = jq.cases.BUILDERS[case]()
    if info.get("golden"):
        g = json.load(open("tests/golden/%s.json" % info["golden"]))
        pcof = np.array(g["pcof0"]) if "pcof0" in g else info["pcof0"]
    else:
        pcof = info["pcof0"]
    p = copy.copy(p)
    p.Integrator_id = jq.Implicit_Midpoint
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=p.N)
    for opt in (None, {"rl_split": 0}):
        wa = jq.Working_Arrays_M_HIP(p, pcof.size, options=opt)
        best = None
        for _ in range(3):
            o = jq.traceobjgrad(pcof, p, wa)
            t = wa.last_timing()
            if best is None or t["ms_forward"] + t["ms_backward"] < best[0] + best[1]:
                best = (t["ms_forward"], t["ms_backward"], t["ms_total"])
        print("%-8s Ntot %2d steps %5d %-14s: fwd %6.3f bwd %6.3f total %6.3f ms  fam %d var %d  objf %.17g" % (case, p.Ntot, p.nsteps, str(opt), best[0], best[1], best[2], t["kernel_family"], t["kernel_variant"], o[0]), flush=True)
        wa.close()
