"""Development aid: the dense cooperative-quad kernels (17 .. 32 levels without the 4 x 4 x n structure) against the cooperative kernels
(option dq=0): one evaluation and ensembles of a random 25-level problem with N = 4 (two five-level subsystems), 2 000 steps."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from oracle.oracle import Oracle
from test_gpu_random import random_problem
rng = np.random.default_rng(2525)
p, pcof = random_problem(jq, rng, 25, 4, 2, 2, 2000, 4, 1, False)
t0 = time.perf_counter()
Oracle(p, use_sparse=False).traceobjgrad(pcof)
tc = time.perf_counter() - t0
print("25 levels, N 4, 2 controls, 2000 steps, m 4; CPU oracle, one core: %.1f ms per evaluation" % (tc * 1e3))
for opts, tag in (({}, "dense cooperative quad"), ({"dq": 0}, "cooperative (dq=0)")):
    wa = jq.Working_Arrays_HIP(p, pcof.size, options=opts)
    msg = "%-24s" % tag
    for ns in (1, 9, 64, 128, 256, 512, 1024):
        x, w = np.polynomial.legendre.leggauss(ns)
        best = None
        for _ in range(2):
            jq.eval_f_g_grad(pcof, p, wa, x * 0.05, w * 0.5, True, shift=0.01 * np.arange(p.Ntot))
            t = wa.last_timing()
            if best is None or t["ms_total"] < best[0]:
                best = (t["ms_total"], t["kernel_family"], t["kernel_band"])
        msg += " | x%4d: %7.2f ms (fam %d)" % (ns, best[0], best[1])
    print(msg, flush=True)
    wa.close()
