"""Development aid: cnot2 ensembles on the dense NT = 1 MFMA slab kernels (usage: JQ_LIB=<variant> python scripts/time_cnot2_ens.py)."""
import json, sys, numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
p, info = jq.cases.BUILDERS["cnot2"]()
g = json.load(open("tests/golden/%s.json" % info["golden"]))
pcof = np.array(g["pcof0"])
wa = jq.Working_Arrays_HIP(p, pcof.size)
for ns in (8192, 32768):
    rng = np.random.default_rng(ns); x, w = rng.uniform(-1, 1, ns), rng.random(ns) / ns      # (leggauss(32768) takes minutes on the host)
    shift = 0.01 * np.arange(p.Ntot)
    best = 1e9
    for r in range(3):
        jq.eval_f_g_grad(pcof, p, wa, x * 0.05, w * 0.5, True, shift=shift)
        t = wa.last_timing(); best = min(best, t["ms_total"])
    print("cnot2 x %d: %.1f ms (fwd %.1f bwd %.1f) family %d infid %.15g" % (ns, best, t["ms_forward"], t["ms_backward"], t["kernel_family"], p.last_infidelity))
