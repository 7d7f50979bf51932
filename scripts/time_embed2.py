"""cnot2 large batches on the embedded twin: quad-layout vs JQ_BW_T4 slab kernels (JQ_QUAD=0)."""
import json, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from conftest import case_inputs
params, info, pcof, _ = case_inputs("cnot2")
for ns in (8192, 16384, 32768, 65536, 131072):
    x, w = np.polynomial.legendre.leggauss(64)
    nodes = np.tile(x, ns // 64) * 0.5 * (2 * np.pi * 2e-2)
    weights = np.tile(w, ns // 64) * 0.5 / (ns // 64)
    shift = 0.05 * np.arange(params.Ntot)
    for env in ({}, {"JQ_QUAD": "0"}, {"JQ_QUAD8": "0"}, {"JQ_QUAD8": "1"}, {"JQ_QUAD8": "2"}):
        os.environ.update(env); os.environ["JQ_EMBED"] = "2"
        wa = jq.Working_Arrays_HIP(params, pcof.size)
        for rep in range(2):
            jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
        for k in list(env) + ["JQ_EMBED"]: os.environ.pop(k, None)
        t = wa.last_timing()
        print("cnot2 x %6d  %-16s family %d band %d  %.1f ms  %.3e SVTS/s" % (ns, env, t["kernel_family"], t["kernel_band"], t["ms_total"], t["svts"] / t["ms_total"] * 1e3), flush=True)
        wa.close()
