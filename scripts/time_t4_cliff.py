import os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from test_gpu_random import random_problem
for Ntot in (96, 112, 128):
    rng = np.random.default_rng(7)
    nsteps = 2000
    p, pcof = random_problem(jq, rng, Ntot, 4, 3, 2, nsteps, 6, 1, "t4")
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    line = "Ntot %3d t4" % Ntot
    for ns in (1, 256, 3072):
        nodes = np.linspace(-1e-3, 1e-3, ns) if ns > 1 else np.zeros(1)
        weights = np.full(ns, 1.0 / ns)
        for rep in range(2):
            jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=np.arange(Ntot) * 1e-3)
        t = wa.last_timing()
        line += "  %4d samples: fam %d <%d,%d> %8.1f ms" % (ns, t["kernel_family"], t["kernel_size"], t["kernel_band"], t["ms_total"])
    print(line, flush=True)
    wa.close()
