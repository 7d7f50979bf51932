"""Experiment: Ntot = 48 (cnot3 with Ng3 = 2: 4 x 4 x 3) large batches: quad-layout plan vs JQ_BW_T4 slab kernels."""
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
params, info = jq.cases.cnot3(Ng3=2)
params.nsteps = 4000
params.T = params.T * 4000 / 32386
rng = np.random.default_rng(3)
pcof = 0.01 * rng.standard_normal(info["nCoeff"])
for ns in (2048, 4096, 8192, 16384):
    nodes = np.linspace(-1e-3, 1e-3, ns); weights = np.full(ns, 1.0 / ns)
    shift = np.arange(params.Ntot) * 1.0
    for env in ({}, {"JQ_QUAD": "0"}):
        os.environ.update(env)
        wa = jq.Working_Arrays_HIP(params, pcof.size)
        for k in env: os.environ.pop(k, None)
        for rep in range(2):
            jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
        t = wa.last_timing()
        print("%s Ntot=%d x %6d  %-16s family %d band %d  %.1f ms  %.3e SVTS/s (x Ntot: %.3e) infid %.10f" % (os.environ.get("JQ_LIB", "default")[-16:], params.Ntot, ns, env, t["kernel_family"], t["kernel_band"],
              t["ms_total"], t["svts"] / t["ms_total"] * 1e3, params.Ntot * t["svts"] / t["ms_total"] * 1e3, params.last_infidelity), flush=True)
        wa.close()
