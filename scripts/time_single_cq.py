"""One cnot3 evaluation on the cooperative-quad kernels: forward / backward sweep times (kernel experiments of the latency path)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juqbox_jl_amd as jq
params, info = jq.cases.cnot3()
pcof = np.array(json.load(open(os.path.join(ROOT, "tests/golden/cnot3.json")))["pcof0"])
wa = jq.Working_Arrays_HIP(params, pcof.size)
best = None
for rep in range(3):
    jq.traceobjgrad(pcof, params, wa, False, True)
    t = wa.last_timing()
    best = t if best is None or t["ms_total"] < best["ms_total"] else best
print("family %d: %.1f ms (fwd %.1f bwd %.1f)" % (best["kernel_family"], best["ms_total"], best["ms_forward"], best["ms_backward"]))
wa.close()
