"""Mid-size ensembles: the backward sweep with the two chains of a column quad on two waves (k_backward_qsplit) next to the one-wave
quad-layout kernel / the two-round cooperative-quad sweep (JQ_QSPLIT=0): times and bit-wise comparison of the results.
python scripts/time_qsplit.py [nsteps] [sizes]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juqbox_jl_amd as jq
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512, 1024, 700, 300]
params, info = jq.cases.cnot3()
if nsteps:
    params.T, params.nsteps = params.T * nsteps / params.nsteps, nsteps
pcof = np.array(json.load(open(os.path.join(ROOT, "tests/golden/cnot3.json")))["pcof0"])
res = {}
for tag, env in (("split", {}), ("one", {"JQ_QSPLIT": "0"})):
    os.environ.update(env)
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    for ns in sizes:
        nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
        best = None
        for rep in range(2):
            jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
            t = wa.last_timing()
            best = t if best is None or t["ms_total"] < best["ms_total"] else best
        res[tag, ns] = (params.last_infidelity, params.last_leak, params.last_infidelity_grad.copy())
        print("%-6s %5d samples: family %d variant %2d  %.1f ms (fwd %.1f bwd %.1f)  infidelity %.15f" %
              (tag, ns, best["kernel_family"], best["reserved"], best["ms_total"], best["ms_forward"], best["ms_backward"], params.last_infidelity), flush=True)
    wa.close()
    for k in env:
        os.environ.pop(k, None)
for ns in sizes:
    a, b = res["split", ns], res["one", ns]
    print("%5d samples: bit-identical %s (infidelity rel diff %.1e, leak rel diff %.1e, gradient rel diff %.1e)" %
          (ns, a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]), abs(a[0] - b[0]) / abs(b[0]), abs(a[1] - b[1]) / max(abs(b[1]), 1e-300),
           np.linalg.norm(a[2] - b[2]) / np.linalg.norm(b[2])))
