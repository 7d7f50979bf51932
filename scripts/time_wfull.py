"""Full leakage weights on the latency path: one cnot3 evaluation (and small ensembles) with REAL forbidden states on the cooperative-quad
kernels next to the quad-layout kernels (JQ_CQ_W=0).  python scripts/time_wfull.py [nsteps] [ranks] [sizes]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juqbox_jl_amd as jq
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ranks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 4]
sizes = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 9, 64]
pcof = np.array(json.load(open(os.path.join(ROOT, "tests/golden/cnot3.json")))["pcof0"])
for r in ranks:
    params, info = jq.cases.cnot3()
    if nsteps:
        params.T, params.nsteps = params.T * nsteps / params.nsteps, nsteps
    if r:
        rng = np.random.default_rng(r)
        fs = rng.standard_normal((params.Ntot, r))
        fs /= np.linalg.norm(fs, axis=0)
        fw = 0.5 + rng.random(r)
        params.wmat_real = np.asfortranarray(sum(fw[k] * np.outer(fs[:, k], fs[:, k]) for k in range(r)))
        params.wmat_imag = np.zeros_like(params.wmat_real)
    res = {}
    for tag, env in (("cq", {}), ("quad", {"JQ_CQ_W": "0"})):
        if r == 0 and tag == "quad":
            continue
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
            print("rank %d %-5s %4d samples: family %d variant %d  %.1f ms (fwd %.1f bwd %.1f)  infidelity %.15f leak %.6e" %
                  (r, tag, ns, best["kernel_family"], best["reserved"], best["ms_total"], best["ms_forward"], best["ms_backward"],
                   params.last_infidelity, params.last_leak), flush=True)
        wa.close()
        for k in env:
            os.environ.pop(k, None)
    if r:
        for ns in sizes:
            a, b = res["cq", ns], res["quad", ns]
            print("rank %d %4d samples: infidelity diff %.1e, leak rel diff %.1e, gradient rel diff %.1e" %
                  (r, ns, abs(a[0] - b[0]), abs(a[1] - b[1]) / abs(b[1]), np.linalg.norm(a[2] - b[2]) / np.linalg.norm(b[2])))
