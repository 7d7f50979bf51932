"""Latency path at other sizes of the 4 x 4 x n structure (cnot3 with fewer cavity levels: NT = 2 .. 6 tile rows) and sample
counts: cooperative-quad kernels vs quad-layout kernels (JQ_CQ=0), 4000 time steps."""
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
for Ng3 in (1, 2, 3, 5):
    params, info = jq.cases.cnot3(Ng3=Ng3)
    params.nsteps = 4000
    params.T = params.T * 4000 / 32386
    rng = np.random.default_rng(3)
    pcof = 0.01 * rng.standard_normal(info["nCoeff"])
    for env in ({}, {"JQ_CQ": "0"}):
        os.environ.update(env)
        wa = jq.Working_Arrays_HIP(params, pcof.size)
        for k in env: os.environ.pop(k, None)
        line = "Ntot=%3d %-14s" % (params.Ntot, env)
        for ns in (1, 256, 512, 768):
            nodes = np.linspace(-1e-3, 1e-3, ns) if ns > 1 else np.zeros(1)
            weights = np.full(ns, 1.0 / ns)
            shift = np.arange(params.Ntot) * 1.0
            for rep in range(2):
                jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
            t = wa.last_timing()
            line += "  %4d: fam %d %6.1f ms" % (ns, t["kernel_family"], t["ms_total"])
        print(line, flush=True)
        wa.close()
