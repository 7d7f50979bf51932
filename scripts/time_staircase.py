"""Time of one cnot3 ensemble evaluation (full length, forward + adjoint) against the ensemble size: the staircase of DESIGN.md
section 7.  python scripts/time_staircase.py [sizes,comma,separated]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import juqbox_jl_amd as jq
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 64, 256, 512, 768, 1024, 2048, 3072, 4096, 8192, 24576]
params, info = jq.cases.cnot3()
pcof = np.array(json.load(open(os.path.join(ROOT, "tests/golden/cnot3.json")))["pcof0"])
wa = jq.Working_Arrays_HIP(params, pcof.size)
for ns in sizes:
    nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
    t0 = time.perf_counter()
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
    dt = time.perf_counter() - t0
    t = wa.last_timing()
    print("%6d samples  %.3f s  %8.1f evals/s  family %d  launches %d+%d" % (ns, dt, ns / dt, t["kernel_family"], t["n_forward_launches"], t["n_backward_launches"]), flush=True)
