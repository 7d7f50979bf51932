"""Rounds of the Stormer-Verlet quad-layout kernels with ONE slab per workgroup (JQ_QUAD8=0): does time scale with the rounds?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import case_inputs
import juqbox_jl_amd as jq

params, info, pcof, _ = case_inputs("cnot3")
os.environ["JQ_QUAD8"] = sys.argv[1] if len(sys.argv) > 1 else "0"
os.environ["JQ_NOSPLIT"] = "1"
wa = jq.Working_Arrays_HIP(params, pcof.size)
for ns in (1024, 2048, 3072, 4096, 5120):
    nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, False, shift=shift)
    t0 = time.perf_counter()
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, False, shift=shift)
    dt = time.perf_counter() - t0
    t = wa.last_timing()
    print("%5d samples  %.3f s  fwd %.1f ms (%d launches) family %d" % (ns, dt, t["ms_forward"], t["n_forward_launches"], t["kernel_family"]), flush=True)
