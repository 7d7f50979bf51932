"""Quick GPU-vs-oracle parity run (development aid; the judged tests live in tests/)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import juqbox_jl_amd as jq  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

cases = sys.argv[1:] or ["rabi", "swap02", "flux", "cnot2"]
for case in cases:
    p, info = jq.cases.BUILDERS[case]()
    if info.get("golden"):
        g = json.load(open("tests/golden/%s.json" % info["golden"]))
        pcof = np.array(g["pcof0"]) if "pcof0" in g else info["pcof0"]
    else:
        pcof = info["pcof0"]
    o = Oracle(p)
    t0 = time.time()
    r = o.traceobjgrad(pcof)
    t_cpu = time.time() - t0
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    t0 = time.time()
    objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, p, wa, False, True)
    t_gpu = time.time() - t0
    tm = wa.last_timing()
    rel = lambda a, b: np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)
    print("%-14s cpu %.2fs gpu %.2fs (dev %.1f ms, prop %.1f ms) | objfv %.3e prim %.3e sec %.3e grad %.3e infidelgrad %.3e"
          % (case, t_cpu, t_gpu, tm["ms_total"], tm["ms_propagate"], rel(objfv, r["objfv"]), rel(prim, r["primaryobjf"]),
             rel(sec, r["secondaryobjf"]), rel(tg, r["totalgrad"]), rel(ig, r["infidelgrad"])), flush=True)
    if p.objFuncType != 1:
        print("   leakgrad", rel(lg, r["leakgrad"]))
