import json, os, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
params, info = jq.cases.cnot3()
pcof = np.array(json.load(open("tests/golden/cnot3.json"))["pcof0"])
nodes, weights, shift = jq.cases.cnot3_ensemble(3072)
out = []
for cs in (None, "5000", "777"):
    if cs: os.environ["JQ_CHUNK_STEPS"] = cs
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    os.environ.pop("JQ_CHUNK_STEPS", None)
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
    t = wa.last_timing()
    out.append((params.last_infidelity, params.last_leak, params.last_infidelity_grad.copy()))
    print(cs, t["kernel_family"], t["kernel_band"], "%.0f ms" % t["ms_total"], "%.15g %.15g" % (params.last_infidelity, params.last_leak))
    wa.close()
for o in out[1:]:
    print(abs(o[0]-out[0][0]), abs(o[1]-out[0][1]), np.linalg.norm(o[2]-out[0][2])/np.linalg.norm(out[0][2]))
