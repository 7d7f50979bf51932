import json, os, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
params, info = jq.cases.cnot3()
params.nsteps = 2000; params.T = params.T * 2000 / 32386
pcof = np.array(json.load(open("tests/golden/cnot3.json"))["pcof0"])
res = {}
for ns in (700, 1500, 3072, 5000):
    nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
    for tag, env in (("plan", {}), ("slab", {"JQ_QUAD": "0", "JQ_COOP_MAX": "0"})):
        os.environ.update(env)
        wa = jq.Working_Arrays_HIP(params, pcof.size)
        for k in env: os.environ.pop(k)
        jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
        t = wa.last_timing()
        res[tag] = (params.last_infidelity, params.last_leak, params.last_infidelity_grad.copy(), t["kernel_family"], t["kernel_band"], t["ms_total"])
        wa.close()
    a, b = res["plan"], res["slab"]
    print("ns=%d plan fam %d band %d %.0f ms | slab %.0f ms | d infid %.1e d leak %.1e d grad %.1e" % (ns, a[3], a[4], a[5], b[5],
          abs(a[0]-b[0])/abs(b[0]), abs(a[1]-b[1])/abs(b[1]), np.linalg.norm(a[2]-b[2])/np.linalg.norm(b[2])))
