"""Cost of full leakage weights (jq_update_wmat: low-rank terms in the kernels) next to the Diagonal fast path.
usage: time_r4_fullw.py [nsteps of cnot3, default: full length]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juqbox_jl_amd as jq  # noqa: E402


def forbid(p, nforb, complex_states, seed=3):
    rng = np.random.default_rng(seed)
    fs = rng.standard_normal((p.Ntot, nforb)) + (1j * rng.standard_normal((p.Ntot, nforb)) if complex_states else 0)
    fs /= np.linalg.norm(fs, axis=0)
    W = sum((0.5 + rng.random()) * np.outer(fs[:, k], fs[:, k].conj()) for k in range(nforb))
    p.wmat_real, p.wmat_imag = np.asfortranarray(W.real.copy()), np.asfortranarray(W.imag.copy())


def timed(case, nforb, complex_states, env=None, nsteps=None, ens=()):
    p, info = jq.cases.BUILDERS[case]()
    g = json.load(open(os.path.join(ROOT, "tests", "golden", info["golden"] + ".json")))
    pcof = np.array(g["pcof0"])
    if nsteps:
        p.T, p.nsteps = p.T * nsteps / p.nsteps, nsteps
    if nforb:
        forbid(p, nforb, complex_states)
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        wa = jq.Working_Arrays_HIP(p, pcof.size)
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)
    best = None
    for _ in range(2):
        jq.traceobjgrad(pcof, p, wa)
        t = wa.last_timing()
        best = t if best is None or t["ms_total"] < best["ms_total"] else best
    msg = "%-7s rank %2d %-7s fam %d <%d,%d>: single %9.2f ms (fwd %8.2f bwd %8.2f)" % (
        case, nforb, "complex" if complex_states else "real", best["kernel_family"], best["kernel_size"], best["kernel_band"], best["ms_total"],
        best["ms_forward"], best["ms_backward"])
    for ns in ens:
        x, w = np.polynomial.legendre.leggauss(ns)
        shift = p.shift_weights_reference() if p.Ntot <= 4 else 0.01 * np.arange(p.Ntot)
        jq.eval_f_g_grad(pcof, p, wa, x * 1e-3, w * 0.5, True, shift=shift)
        t = wa.last_timing()
        msg += " | x%d fam %d: %9.2f ms" % (ns, t["kernel_family"], t["ms_total"])
    print(msg, flush=True)
    wa.close()


ns3 = int(sys.argv[1]) if len(sys.argv) > 1 else None
print("library:", jq._lib.load().jq_version().decode())
for r, cplx in ((0, False), (2, False), (2, True), (8, True)):
    timed("swap02", r, cplx, ens=(512,))
for r, cplx in ((0, False), (2, True)):
    timed("cnot2", r, cplx, ens=(512, 8192))
timed("cnot3", 0, False, nsteps=ns3, ens=(512,))
timed("cnot3", 0, False, {"JQ_CQ": "0"}, nsteps=ns3, ens=(512,))
for r, cplx in ((1, False), (1, True), (2, True), (4, True), (8, True)):
    timed("cnot3", r, cplx, nsteps=ns3, ens=(512,) if r in (2, 8) else ())
timed("cnot3", 2, True, {"JQ_T4": "0"}, nsteps=4000)
timed("cnot3", 0, False, {"JQ_T4": "0"}, nsteps=4000)
timed("cnot3", 2, True, {"JQ_T4": "0", "JQ_COOP_MAX": "0"}, nsteps=4000)
timed("cnot3", 0, False, {"JQ_T4": "0", "JQ_COOP_MAX": "0"}, nsteps=4000)
