"""Experiment: per-interval cycle counts of one backward step of the cooperative-quad kernels (library built with -DJQ_CQ_TIMING)."""
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
params, info = jq.cases.cnot3()
params.nsteps = 1000
params.T = params.T * 1000 / 32386
pcof = np.array(json.load(open("tests/golden/cnot3.json"))["pcof0"])
wa = jq.Working_Arrays_HIP(params, pcof.size)
jq.eval_f_g_grad(pcof, params, wa, np.zeros(1), np.ones(1), True)
print(wa.last_timing())
