"""Profiling target: three single evaluations of an unstructured 32-level problem (2 000 steps) -- the cooperative kernels in latency mode."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from test_gpu_random import random_problem
Ntot = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(100 + Ntot)
p, pcof = random_problem(jq, rng, Ntot, 4, 2, 2, 2000, 4, 1, False)
wa = jq.Working_Arrays_HIP(p, pcof.size)
for _ in range(3):
    jq.traceobjgrad(pcof, p, wa)
print(wa.last_timing())
wa.close()
