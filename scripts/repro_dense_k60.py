"""Slab kernels <6, 0> / <6, 5> (dense 96 x 96 operators) against the oracle for short runs with odd and even numbers of steps --
the check that found the miscompiled full-weights object w_6_5 (VGPR register form, odd chunk lengths) in round 5."""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq
from oracle.oracle import Oracle
from test_gpu_random import random_problem
worst = 0.0
for structure, env, wts in ((False, {"JQ_COOP_MAX": "0"}, False), (False, {"JQ_COOP_MAX": "0"}, True), ("t4", {"JQ_FORCE_DENSE": "1", "JQ_EMBED": "0"}, False),
                            ("t4", {"JQ_FORCE_DENSE": "1", "JQ_EMBED": "0"}, True), (True, {"JQ_COOP_MAX": "0", "JQ_OD": "0"}, False), (True, {"JQ_COOP_MAX": "0", "JQ_OD": "0"}, True),
                            (False, {"JQ_ROWLANE_MAX": "0", "JQ_LANE": "0"}, True), ("t4", {"JQ_CQ": "0"}, True), ("t4", {"JQ_CQ": "0", "JQ_QUAD": "0"}, True)):
    for Ntot in ((12, 16) if "JQ_ROWLANE_MAX" in env else (96, 80, 33) if "JQ_CQ" not in env else (96, 64, 32)):
        for ns in (3, 4, 5, 8):
            for m in (0, 3, 6):
                rng = np.random.default_rng(ns + 10 * m)
                p, pcof = random_problem(jq, rng, Ntot, 4, 2, 1, ns, m, 3, structure)
                if wts:
                    fs = rng.standard_normal((Ntot, 2)) + 1j * rng.standard_normal((Ntot, 2))
                    fs = fs / np.linalg.norm(fs, axis=0)
                    W = sum((0.5 + 0.3 * k) * np.outer(fs[:, k], np.conj(fs[:, k])) for k in range(2))
                    p.wmat_real, p.wmat_imag = np.asfortranarray(W.real.copy()), np.asfortranarray(W.imag.copy())
                wa = jq.Working_Arrays_HIP(p, pcof.size, options=env)      # (historic spelling: {"JQ_QUAD": "0"} = option quad=0)
                r = Oracle(p, use_sparse=False).traceobjgrad(pcof)
                try:
                    out = jq.traceobjgrad(pcof, p, wa, False, True)
                except RuntimeError as e:      # (a combination without kernels: refused, never evaluated otherwise)
                    print("%-5s Ntot %3d weights %-5s steps %d m %d: refused (%s)" % (structure, Ntot, wts, ns, m, str(e)[-60:]), flush=True)
                    wa.close()
                    continue
                t = wa.last_timing()
                e = max(abs(out[2] - r["primaryobjf"]) / abs(r["primaryobjf"]), abs(out[3] - r["secondaryobjf"]) / abs(r["secondaryobjf"]),
                        np.linalg.norm(out[1] - r["totalgrad"]) / np.linalg.norm(r["totalgrad"]))
                worst = max(worst, e)
                print("%-5s Ntot %3d weights %-5s steps %d m %d: family %d <%d, %d>  err %.1e%s" % (structure, Ntot, wts, ns, m, t["kernel_family"], t["kernel_size"], t["kernel_band"], e,
                      "   <<<<<< MISMATCH" if e > 1e-9 else ""), flush=True)
                wa.close()
print("worst %.2e" % worst)
