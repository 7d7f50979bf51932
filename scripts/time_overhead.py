"""Per-call host overhead of small evaluations: wall time of jq.traceobjgrad vs the device time of its kernels."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from conftest import case_inputs
for case in ("rabi", "swap02", "cnot2"):
    params, info, pcof, g = case_inputs(case)
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    for _ in range(3): jq.traceobjgrad(pcof, params, wa, False, True)
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): jq.traceobjgrad(pcof, params, wa, False, True)
    wall = (time.perf_counter() - t0) / n * 1e3
    t = wa.last_timing()
    print("%-7s wall %.3f ms per call; device total %.3f ms (propagators %.3f ms, %d + %d launches)" % (case, wall, t["ms_total"], t["ms_propagate"], t["n_forward_launches"], t["n_backward_launches"]), flush=True)
    wa.close()
