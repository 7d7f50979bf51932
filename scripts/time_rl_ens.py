"""Development aid: forward / backward time of mid-size ensembles on the row-lane kernels (library = JQ_LIB)."""
import json, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [256, 384, 512, 768, 1024]
for case in ["swap02_rn", "cnot1", "cnot2"]:
    p, info = jq.cases.BUILDERS[case]()
    if info.get("golden"):
        g = json.load(open("tests/golden/%s.json" % info["golden"]))
        pcof = np.array(g["pcof0"]) if "pcof0" in g else info["pcof0"]
    else:
        pcof = info["pcof0"]
    shift = p.shift_weights_reference() if p.Ntot <= 4 else 0.01 * np.arange(p.Ntot)
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    msg = "%-9s Ntot %2d:" % (case, p.Ntot)
    for ns in sizes:
        x, w = np.polynomial.legendre.leggauss(ns)
        best = None
        for _ in range(3):
            jq.eval_f_g_grad(pcof, p, wa, x * 0.05, w * 0.5, True, shift=shift)
            t = wa.last_timing()
            if best is None or t["ms_total"] < best[2]:
                best = (t["ms_forward"], t["ms_backward"], t["ms_total"])
        msg += "  x%5d: fwd %5.2f bwd %5.2f total %5.2f" % (ns, *best)
    print(msg, flush=True)
    wa.close()
