"""Large ensembles (size-independent property: all nodes equal -> the ensemble result is the single evaluation times sum(w))."""
import json, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from conftest import case_inputs
for case, ns in (("swap02", 1_000_000), ("cnot2", 262_144), ("cnot3", 20_000)):
    params, info, pcof, g = case_inputs(case)
    if case == "cnot3":
        params.nsteps = 2000
        params.T = params.T * 2000 / 32386
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    jq.eval_f_g_grad(pcof, params, wa, np.array([1e-3]), np.ones(1), True, shift=np.arange(params.Ntot) * 0.01)
    f1, g1 = params.last_infidelity, params.last_infidelity_grad.copy()
    rng = np.random.default_rng(1)
    w = rng.random(ns)
    t0 = time.perf_counter()
    jq.eval_f_g_grad(pcof, params, wa, np.full(ns, 1e-3), w, True, shift=np.arange(params.Ntot) * 0.01)
    el = time.perf_counter() - t0
    t = wa.last_timing()
    print("%-7s %8d samples: %.2f s (device %.0f ms, family %d)  infid/(f1 sum w) - 1 = %.1e   grad rel %.1e" % (
        case, ns, el, t["ms_total"], t["kernel_family"], params.last_infidelity / (f1 * w.sum()) - 1.0,
        np.linalg.norm(params.last_infidelity_grad - g1 * w.sum()) / np.linalg.norm(g1 * w.sum())), flush=True)
    wa.close()
