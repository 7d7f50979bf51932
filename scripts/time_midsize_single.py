"""Development aid: ONE evaluation (the Ipopt caller's case) of random problems between the row-lane kernels' 16 levels and cnot3's 96:
which kernel family serves them, microseconds per time step, next to the CPU oracle on one core."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from oracle.oracle import Oracle
from test_gpu_random import random_problem
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for Ntot, N, banded in [(20, 4, False), (32, 4, False), (32, 4, True), (48, 4, False), (64, 8, False), (64, 4, "t4"), (96, 4, False), (96, 4, True), (128, 4, False)]:
    rng = np.random.default_rng(100 + Ntot)
    p, pcof = random_problem(jq, rng, Ntot, N, 2, 2, nsteps, 4, 1, banded)
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    best = None
    for _ in range(3):
        jq.traceobjgrad(pcof, p, wa)
        t = wa.last_timing()
        if best is None or t["ms_total"] < best["ms_total"]:
            best = dict(t)
    t0 = time.perf_counter()
    Oracle(p, use_sparse=False).traceobjgrad(pcof)
    tc = time.perf_counter() - t0
    print("Ntot %3d N %d %-5s: family %d <%d,%d> var %d  fwd %7.2f bwd %7.2f total %7.2f ms = %5.2f us/step | CPU oracle %7.1f ms (%.1f x)" % (
        Ntot, N, banded, best["kernel_family"], best["kernel_size"], best["kernel_band"], best["kernel_variant"], best["ms_forward"], best["ms_backward"],
        best["ms_total"], best["ms_total"] * 1e3 / nsteps, tc * 1e3, tc * 1e3 / best["ms_total"]), flush=True)
    wa.close()
