"""Ensemble sizes between full rounds of the three-slab quad-layout kernels (cnot3): split batches (default) vs JQ_NOSPLIT=1."""
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
params, info = jq.cases.cnot3()
pcof = np.array(json.load(open("tests/golden/cnot3.json"))["pcof0"])
wa = jq.Working_Arrays_HIP(params, pcof.size)
for ns in (300, 600, 1100, 1500, 2200, 2600, 3200, 4096, 5000):
    n2, w2, s2 = jq.cases.cnot3_ensemble(ns)
    line = "%5d samples:" % ns
    ref = None
    for env in ({}, {"JQ_NOSPLIT": "1"}):
        os.environ.update(env)
        jq.eval_f_g_grad(pcof, params, wa, n2, w2, True, shift=s2)
        for k in env: os.environ.pop(k, None)
        t = wa.last_timing()
        g = params.last_infidelity_grad.copy()
        if ref is None: ref = (params.last_infidelity, g)
        line += "  %-18s %7.1f ms = %5.0f evals/s" % (env or "split", t["ms_total"], ns / t["ms_total"] * 1e3)
    line += "  |d infid| %.1e  |d grad|/|grad| %.1e" % (abs(params.last_infidelity - ref[0]), np.linalg.norm(g - ref[1]) / np.linalg.norm(g))
    print(line, flush=True)
