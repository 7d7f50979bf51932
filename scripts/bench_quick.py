"""Development timing helper: one cnot3 ensemble evaluation, prints fwd/bwd kernel times."""
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
noshift = len(sys.argv) > 2 and sys.argv[2] == "noshift"
params, info = jq.cases.cnot3()
pcof = np.array(json.load(open("tests/golden/cnot3.json"))["pcof0"])
nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
if noshift:
    nodes = np.zeros_like(nodes)
wa = jq.Working_Arrays_HIP(params, pcof.size)
jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
t = wa.last_timing()
print("JQ_DEBUG=%s noshift=%s: total %.0f ms fwd %.0f bwd %.0f -> %.1f evals/s ; infid %.12f" % (
    os.environ.get("JQ_DEBUG", "0"), noshift, t["ms_total"], t["ms_forward"], t["ms_backward"], ns / t["ms_total"] * 1e3, params.last_infidelity))
