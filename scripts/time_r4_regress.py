"""Regression timing of the Diagonal fast paths across library builds (round 4: the kernels gained the low-rank full leakage weights).
usage: time_r4_regress.py [repo root to import the package from]   -- one line per (case, kernel family), ms per evaluation.
Run it once per build (the current tree, a copy of an older one under juqbox.jl_amd/exp/<tag>/) and compare the columns."""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juqbox_jl_amd as jq  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ENVS = ("JQ_COOP_MAX", "JQ_LANE", "JQ_ROWLANE_MAX", "JQ_T4", "JQ_OD", "JQ_QUAD", "JQ_CQ", "JQ_EMBED", "JQ_FORCE_DENSE", "JQ_RL_SPLIT", "JQ_QUAD8")


def best_of(f, wa, reps=3):
    best = None
    for _ in range(reps):
        f()
        t = wa.last_timing()
        if best is None or t["ms_total"] < best["ms_total"]:
            best = t
    return best


def run(case, label, env=None, nsteps=None, ens=()):
    p, info = jq.cases.BUILDERS[case]()
    if info.get("golden"):
        g = json.load(open(os.path.join(GOLD, info["golden"] + ".json")))
        pcof = np.array(g["pcof0"]) if "pcof0" in g else np.asarray(info["pcof0"])
    else:
        pcof = np.asarray(info["pcof0"])
    if nsteps:
        p.T, p.nsteps = p.T * nsteps / p.nsteps, nsteps
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        wa = jq.Working_Arrays_HIP(p, pcof.size)
        t = best_of(lambda: jq.traceobjgrad(pcof, p, wa), wa)
        msg = "%-10s %-22s fam %d <%d,%d> single %9.3f ms (fwd %.3f bwd %.3f)" % (case, label, t["kernel_family"], t["kernel_size"], t["kernel_band"],
                                                                               t["ms_total"], t["ms_forward"], t["ms_backward"])
        for ns in ens:
            x, w = np.polynomial.legendre.leggauss(ns)
            shift = p.shift_weights_reference() if p.Ntot <= 4 else 0.01 * np.arange(p.Ntot)
            t = best_of(lambda: jq.eval_f_g_grad(pcof, p, wa, x * 1e-3, w * 0.5, True, shift=shift), wa, 2)
            msg += " | x%d fam %d: %9.3f ms" % (ns, t["kernel_family"], t["ms_total"])
        print(msg, flush=True)
        wa.close()
    finally:
        for k in ENVS:
            os.environ.pop(k, None)


print("library:", jq._lib.load().jq_version().decode())
for case in ("swap02", "flux", "cnot1", "cnot2"):
    run(case, "auto", ens=(512, 8192))
run("swap02", "rowlane one wave", {"JQ_RL_SPLIT": "0"})
run("cnot2", "rowlane one wave", {"JQ_RL_SPLIT": "0"})
run("cnot2", "slab <1,0>", {"JQ_LANE": "0", "JQ_EMBED": "0"}, ens=(8192,))
S = 4000
run("cnot3", "auto (cq)", nsteps=S, ens=(512,))
run("cnot3", "quad", {"JQ_CQ": "0"}, nsteps=S, ens=(1024, 3072))
run("cnot3", "coop od", {"JQ_T4": "0"}, nsteps=S)
run("cnot3", "slab od", {"JQ_T4": "0", "JQ_COOP_MAX": "0"}, nsteps=S, ens=(1024,))
run("cnot3", "coop band", {"JQ_T4": "0", "JQ_OD": "0"}, nsteps=S)
run("cnot3", "slab band", {"JQ_T4": "0", "JQ_OD": "0", "JQ_COOP_MAX": "0"}, nsteps=S, ens=(1024,))
run("cnot3", "slab t4", {"JQ_QUAD": "0", "JQ_COOP_MAX": "0"}, nsteps=S, ens=(4096,))
