"""Cost of creating a handle (Working_Arrays_HIP) and of the first evaluation on it (the reference's plot_results allocates fresh
working arrays per call, src/plot-results.jl:29)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import case_inputs
import juqbox_jl_amd as jq
for case in ("swap02", "cnot2", "cnot3"):
    params, info, pcof, _ = case_inputs(case)
    for rep in range(3):
        t0 = time.perf_counter(); wa = jq.Working_Arrays_HIP(params, pcof.size); t1 = time.perf_counter()
        jq.traceobjgrad(pcof, params, wa, False, True); t2 = time.perf_counter()
        jq.traceobjgrad(pcof, params, wa, False, True); t3 = time.perf_counter()
        wa.close(); t4 = time.perf_counter()
        print("%-8s create %.1f ms  first eval %.1f ms  second eval %.1f ms  destroy %.1f ms" % (case, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
