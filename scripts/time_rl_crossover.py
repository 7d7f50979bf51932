"""Row-lane (two-wave backward sweep) against lane kernels around the cross-over batch size (SWAP-02 risk-neutral, cnot2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import case_inputs
import juqbox_jl_amd as jq
for case in ("swap02_rn", "cnot2"):
    for env in ({}, {"JQ_ROWLANE_MAX": "1000000"}, {"JQ_ROWLANE_MAX": "0"}, {"JQ_RL_SPLIT": "0", "JQ_ROWLANE_MAX": "1000000"}):
        os.environ.update(env)
        params, info, pcof, _ = case_inputs(case)
        wa = jq.Working_Arrays_HIP(params, pcof.size)
        out = []
        for ns in (256, 512, 1024, 2048, 4096, 8192):
            x, w = np.polynomial.legendre.leggauss(min(ns, 2048))
            nodes = np.resize(x, ns) * 0.5 * 2 * np.pi * 2e-2
            weights = np.resize(w, ns) * 0.5
            shift = params.shift_weights_reference() if params.Ntot <= 4 else 0.05 * np.arange(params.Ntot)
            jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
            jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
            t = wa.last_timing()
            out.append("%d: %.1f ms (f%d)" % (ns, t["ms_total"], t["kernel_family"]))
        print("%-10s %-50s %s" % (case, env, "  ".join(out)), flush=True)
        wa.close()
        for k in env:
            os.environ.pop(k, None)
