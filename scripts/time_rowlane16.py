"""A/B aid: single evaluation and small ensembles of random Ntot = 14 / 16 problems (row-lane kernels with NPJ = 16, the instantiation with the
most registers) -- python scripts/time_rowlane16.py  (JQ_LIB selects the build)"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import juqbox_jl_amd as jq
from test_gpu_random import random_problem
for Ntot, N, oft in ((16, 4, 1), (14, 3, 3), (16, 16, 1)):
    p, pcof = random_problem(jq, np.random.default_rng(Ntot + N), Ntot, N, 2, 2, 4000, 4, oft, False)
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    for ns in (1, 64):
        nodes, weights = 0.01 * np.linspace(-1, 1, ns), np.full(ns, 1.0 / ns)
        for _ in range(2):
            jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=np.linspace(0, 1, Ntot))
        t = wa.last_timing()
        print("Ntot %2d N %2d oft %d x %3d: family %d <%d> variant %2d  fwd %7.3f bwd %7.3f ms  infid %.17g" % (Ntot, N, oft, ns, t["kernel_family"], t["kernel_size"], t["kernel_variant"], t["ms_forward"], t["ms_backward"], p.last_infidelity), flush=True)
    wa.close()
