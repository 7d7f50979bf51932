"""Development aid: run a few evaluations of one configuration (for rocprofv3).  usage: prof_case.py case [nsamples] [reps] [imr]"""
import json, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
case = sys.argv[1]
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
p, info = jq.cases.BUILDERS[case]()
if info.get("golden"):
    g = json.load(open("tests/golden/%s.json" % info["golden"]))
    pcof = np.array(g["pcof0"]) if "pcof0" in g else info["pcof0"]
else:
    pcof = info["pcof0"]
imr = len(sys.argv) > 4 and sys.argv[4] == "imr"
if imr:
    p.Integrator_id = jq.Implicit_Midpoint
    p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=p.N)
wa = (jq.Working_Arrays_M_HIP if imr else jq.Working_Arrays_HIP)(p, pcof.size)
x, w = np.polynomial.legendre.leggauss(ns)
shift = p.shift_weights_reference() if p.Ntot <= 4 else 0.01 * np.arange(p.Ntot)
for _ in range(reps):
    if ns == 1:
        jq.traceobjgrad(pcof, p, wa)
    else:
        jq.eval_f_g_grad(pcof, p, wa, x * 0.05, w * 0.5, True, shift=shift)
    print(wa.last_timing())
