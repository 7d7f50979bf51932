"""Development aid: forward / backward time of single evaluations of the small reference cases (row-lane kernels), library = JQ_LIB."""
import json, sys
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
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    best = None
    for _ in range(4):
        o = jq.traceobjgrad(pcof, p, wa)
        t = wa.last_timing()
        if best is None or t["ms_forward"] + t["ms_backward"] < best[0] + best[1]:
            best = (t["ms_forward"], t["ms_backward"], t["ms_total"])
    print("%-8s Ntot %2d m %2d steps %5d: fwd %6.3f bwd %6.3f total %6.3f ms  var %d  objf %.17g" % (case, p.Ntot, p.linear_solver.max_iter, p.nsteps, best[0], best[1], best[2], t["kernel_variant"], o[0]), flush=True)
    wa.close()
