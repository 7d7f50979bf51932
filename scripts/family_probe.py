import json, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
for case in ("cnot2", "cnot1", "flux"):
    p, info = jq.cases.BUILDERS[case]()
    g = json.load(open("tests/golden/%s.json" % info["golden"])) if info.get("golden") else None
    pcof = np.array(g["pcof0"]) if g and "pcof0" in g else info["pcof0"]
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    for ns in (4096, 16384):
        x, w = np.linspace(-1, 1, ns), np.full(ns, 2.0 / ns)     # (leggauss(ns) is an O(ns^3) eigenvalue problem)
        jq.eval_f_g_grad(pcof, p, wa, x * 0.05, w * 0.5, True, shift=0.01 * np.arange(p.Ntot))
        t = wa.last_timing()
        print(case, "Ntot", p.Ntot, "N", p.N, "ns", ns, "family", t["kernel_family"], "size", t["kernel_size"], "band", t["kernel_band"], "%.1f ms" % t["ms_total"], "%.3g SVTS/s" % (ns * p.N * p.nsteps / t["ms_total"] * 1e3))
    wa.close()
