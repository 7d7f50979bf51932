import os, sys, json
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from oracle.oracle import Oracle
from test_gpu_random import random_problem
def run(Nc, nsteps, m, dense_scale, cplx, nf, label, env=None):
    rng = np.random.default_rng(5)
    p, pcof = random_problem(jq, rng, 96, 4, Nc, 1, nsteps, m, 1, "t4")
    D = rng.standard_normal((96, 96))
    p.Hconst = p.Hconst + dense_scale * (D + D.T)
    fs = rng.standard_normal((96, nf)) + (1j * rng.standard_normal((96, nf)) if cplx else 0)
    fs = fs / np.linalg.norm(fs, axis=0)
    W = sum((0.5 + 0.3 * k) * np.outer(fs[:, k], np.conj(fs[:, k])) for k in range(nf))
    p.wmat_real, p.wmat_imag = np.asfortranarray(W.real.copy()), np.asfortranarray(W.imag.copy())
    os.environ.update(env or {})
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    for k in (env or {}): os.environ.pop(k)
    r = Oracle(p, use_sparse=False).traceobjgrad(pcof)
    out = jq.traceobjgrad(pcof, p, wa, False, True)
    t = wa.last_timing()
    gn = np.linalg.norm(r["totalgrad"])
    print("%-40s fam %d band %d: primary %.3e leak %.3e grad %.3e" % (label, t["kernel_family"], t["kernel_band"],
          abs(out[2] - r["primaryobjf"]) / abs(r["primaryobjf"]), abs(out[3] - r["secondaryobjf"]) / abs(r["secondaryobjf"]),
          np.linalg.norm(out[1] - r["totalgrad"]) / gn), flush=True)
    wa.close()
for ns in (3, 4, 5, 6, 8, 12, 16, 24):
    run(1, ns, 6, 0.02, True, 2, "steps %d" % ns)
for ns in (3, 6):
    run(1, ns, 6, 0.02, True, 2, "steps %d JQ_WINDOW=0" % ns, {"JQ_WINDOW": "0"})
    run(1, ns, 6, 0.02, True, 2, "steps %d JQ_CHUNK_STEPS=1" % ns, {"JQ_CHUNK_STEPS": "1"})
    run(1, ns, 0, 0.02, True, 2, "steps %d m=0" % ns)
