"""Ntot > 96 (cooperative kernels with the operator tiles read from HBM / L2): time per evaluation vs the CPU oracle (one core),
single evaluations and ensembles, dense and block-banded operators."""
import os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from test_gpu_random import random_problem
from oracle.oracle import Oracle
for Ntot, banded in ((128, False), (128, True), (192, False), (256, False), (256, True)):
    rng = np.random.default_rng(7)
    nsteps = 2000
    p, pcof = random_problem(jq, rng, Ntot, 4, 2, 2, nsteps, 4, 1, banded)
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    line = "Ntot %3d %-6s" % (Ntot, "band" if banded else "dense")
    for ns in (1, 64, 1024):
        nodes = np.linspace(-1e-3, 1e-3, ns) if ns > 1 else np.zeros(1)
        weights = np.full(ns, 1.0 / ns)
        for rep in range(2):
            jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=np.arange(Ntot) * 1e-3)
        t = wa.last_timing()
        line += "  %4d samples: fam %d <%d,%d> %8.1f ms" % (ns, t["kernel_family"], t["kernel_size"], t["kernel_band"], t["ms_total"])
    p2, _ = random_problem(jq, np.random.default_rng(7), Ntot, 4, 2, 2, 200, 4, 1, banded)
    t0 = time.perf_counter()
    Oracle(p2).traceobjgrad(pcof)
    cpu = (time.perf_counter() - t0) * nsteps / 200
    print(line + "   CPU oracle (1 core, sparse products): %.0f ms per evaluation" % (cpu * 1e3), flush=True)
    wa.close()
