"""Experiment: throughput of the slab kernels on a Kronecker-structured Ntot = 48 problem (NT = 3).
usage: bench_od48.py [nsamples] [od|t4]"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from test_gpu_random import random_problem
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(5)
kind = sys.argv[2] if len(sys.argv) > 2 else "od"
p, pcof = random_problem(jq, rng, 48, 4, 3, 1, 2000, 6, 1, kind)
wa = jq.Working_Arrays_HIP(p, pcof.size)
x, w = np.polynomial.legendre.leggauss(ns)
shift = 0.01 * np.arange(48.0)
for _ in range(2):
    jq.eval_f_g_grad(pcof, p, wa, 0.05 * x, 0.5 * w, True, shift=shift)
    t = wa.last_timing()
    print("ns=%d fam=%d<%d,%d> total %.1f ms fwd %.1f bwd %.1f" % (ns, t["kernel_family"], t["kernel_size"], t["kernel_band"], t["ms_total"], t["ms_forward"], t["ms_backward"]))
