"""cnot2 (test/cases/cnot2-setup.jl, Ntot = 12 = 3 x 4) large-batch throughput with and without the structure embedding
(JQ_EMBED=0: dense NT = 1 MFMA slab kernels; default: embedded twin on the quad-layout / JQ_BW_T4 kernels)."""
import json, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from conftest import case_inputs
params, info, pcof, _ = case_inputs("cnot2")
for ns in (4096, 16384, 65536):
    x, w = np.polynomial.legendre.leggauss(64)
    nodes = np.tile(x, ns // 64) * 0.5 * (2 * np.pi * 2e-2)
    weights = np.tile(w, ns // 64) * 0.5 / (ns // 64)
    shift = 0.05 * np.arange(params.Ntot)
    for mode in ("0", "1"):
        os.environ["JQ_EMBED"] = mode
        wa = jq.Working_Arrays_HIP(params, pcof.size)
        os.environ.pop("JQ_EMBED")
        for rep in range(2):
            jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
        t = wa.last_timing()
        print("cnot2 x %6d samples  JQ_EMBED=%s: family %d band %d  %.1f ms  %.3e SVTS/s  infid %.12f" % (
            ns, mode, t["kernel_family"], t["kernel_band"], t["ms_total"], t["svts"] / t["ms_total"] * 1e3, params.last_infidelity), flush=True)
        wa.close()
