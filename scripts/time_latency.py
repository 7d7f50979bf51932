"""Latency path: one cnot3 evaluation and small ensembles, cooperative-quad kernels vs quad-layout kernels (JQ_CQ=0)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
params, info = jq.cases.cnot3()
pcof = np.array(json.load(open("tests/golden/cnot3.json"))["pcof0"])
g = json.load(open("tests/golden/cnot3.json"))
for env in ({}, {"JQ_CQ": "0"}):
    os.environ.update(env)
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    for k in env: os.environ.pop(k, None)
    for ns in (1, 9, 64, 256, 257, 1024):
        nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
        if ns == 1: nodes = np.zeros(1); weights = np.ones(1)
        for rep in range(2):
            t0 = time.perf_counter()
            jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
            el = time.perf_counter() - t0
        t = wa.last_timing()
        extra = ""
        if ns == 1:
            gt = params.last_infidelity_grad + jq.setup_utils.tikhonov_grad(pcof, params.tik0)
            extra = " golden grad rel err %.1e" % (np.linalg.norm(gt - np.array(g["grad0"])) / np.linalg.norm(g["grad0"]))
        print("%-14s %5d samples: family %d  %.1f ms (fwd %.1f bwd %.1f) wall %.3f s  infid %.12f%s" % (env, ns, t["kernel_family"], t["ms_total"], t["ms_forward"], t["ms_backward"], el, params.last_infidelity, extra), flush=True)
    wa.close()
