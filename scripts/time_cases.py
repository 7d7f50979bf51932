"""Development aid: single-sample and ensemble timings of every configuration (GPU)."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
for case in ["rabi", "swap02", "flux", "cnot1", "cnot2", "cnot3", "swap02_rn"]:
    p, info = jq.cases.BUILDERS[case]()
    if info.get("golden"):
        g = json.load(open("tests/golden/%s.json" % info["golden"]))
        pcof = np.array(g["pcof0"]) if "pcof0" in g else info["pcof0"]
    else:
        pcof = info["pcof0"]
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    jq.traceobjgrad(pcof, p, wa)
    t = wa.last_timing()
    msg = "%-10s Ntot=%3d nsteps=%6d single: %8.1f ms (fwd %.1f bwd %.1f)" % (case, p.Ntot, p.nsteps, t["ms_total"], t["ms_forward"], t["ms_backward"])
    for ns in (512, 8192):
        if case == "cnot3" and ns > 512:
            continue
        x, w = np.polynomial.legendre.leggauss(ns)
        shift = p.shift_weights_reference() if p.Ntot <= 4 else 0.01 * np.arange(p.Ntot)
        jq.eval_f_g_grad(pcof, p, wa, x * 0.05, w * 0.5, True, shift=shift)
        t = wa.last_timing()
        msg += " | ens%d: %8.1f ms = %.3g SVTS/s" % (ns, t["ms_total"], t["svts"] / t["ms_total"] * 1e3)
    print(msg, flush=True)
    wa.close()
