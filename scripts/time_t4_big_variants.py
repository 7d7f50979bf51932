"""4 x 4 x 7 / 4 x 4 x 8 problems (Ntot 112, 128): the quad-layout variants with one / two / three slabs per workgroup and the plan's
own choice, per batch size.  python scripts/time_t4_big_variants.py"""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from test_gpu_random import random_problem
for Ntot in (112, 128):
    rng = np.random.default_rng(7)
    p, pcof = random_problem(jq, rng, Ntot, 4, 3, 2, 2000, 6, 1, "t4")
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    for ns in (1024, 2048, 3072, 6144):
        nodes, weights = np.linspace(-1e-3, 1e-3, ns), np.full(ns, 1.0 / ns)
        line = "Ntot %3d %5d samples:" % (Ntot, ns)
        for tag, env in (("plan", {}), ("1 slab", {"JQ_QUAD8": "0"}), ("2 slabs", {"JQ_QUAD8": "1"}), ("3 slabs", {"JQ_QUAD8": "2"}), ("slab kernels", {"JQ_QUAD": "0"})):
            os.environ.update(env)
            try:
                if tag == "slab kernels":
                    wb = jq.Working_Arrays_HIP(p, pcof.size)
                else:
                    wb = wa
                for rep in range(2):
                    jq.eval_f_g_grad(pcof, p, wb, nodes, weights, True, shift=np.arange(Ntot) * 1e-3)
                t = wb.last_timing()
                line += "  %s %.0f ms (fam %d)" % (tag, t["ms_total"], t["kernel_family"])
                if wb is not wa:
                    wb.close()
            finally:
                for k in env:
                    os.environ.pop(k, None)
        print(line, flush=True)
    wa.close()
