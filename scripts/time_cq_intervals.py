"""Latency path: what a publication interval of the cooperative-quad kernels costs.  The number of Neumann terms m only changes
the number of Horner publications (2 m per forward step, 2 m per backward step and chain), so the slope of the time in m is
the cost of one Horner interval and the intercept the cost of the other six (results with m != 6 are not the golden's)."""
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
params, info = jq.cases.cnot3()
pcof = np.array(json.load(open("tests/golden/cnot3.json"))["pcof0"])
wa = jq.Working_Arrays_HIP(params, pcof.size)
nodes = np.zeros(1); weights = np.ones(1)
res = {}
for m in (6, 4, 2, 1, 0):
    params.linear_solver.max_iter = m
    for rep in range(2):
        jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True)
    t = wa.last_timing()
    res[m] = (t["ms_forward"], t["ms_backward"])
    print("m = %d: family %d fwd %.1f ms bwd %.1f ms" % (m, t["kernel_family"], t["ms_forward"], t["ms_backward"]), flush=True)
ns = params.nsteps
for name, k, nh in (("forward", 0, 2), ("backward", 1, 2)):
    slope = (res[6][k] - res[2][k]) / 4 / nh
    print("%s: %.0f clk per Horner interval, %.0f clk per step for everything else (2.4 GHz)" % (name, slope * 1e-3 / ns * 2.4e9, (res[6][k] - 6 * nh * slope) * 1e-3 / ns * 2.4e9))
