"""Development aid: backward sweep of the row-lane kernels on one, two and three waves per four columns (option rl_split = 0 / 2 / 3)
over the ensemble size -- the measurement behind the occupancy rule of jq_host_eval.h (JQ_RL_ROOM)."""
import json, sys
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 64, 256, 512, 768, 1024, 1536, 2048]
for case in ["swap02", "cnot1", "cnot2"]:
    p, info = jq.cases.BUILDERS[case]()
    if info.get("golden"):
        g = json.load(open("tests/golden/%s.json" % info["golden"]))
        pcof = np.array(g["pcof0"]) if "pcof0" in g else info["pcof0"]
    else:
        pcof = info["pcof0"]
    shift = p.shift_weights_reference() if p.Ntot <= 4 else 0.01 * np.arange(p.Ntot)
    for ns in sizes:
        msg = "%-7s Ntot %2d x %5d:" % (case, p.Ntot, ns)
        for split in (0, 2, 3, 1):
            wa = jq.Working_Arrays_HIP(p, pcof.size, options={"rl_split": split})
            x, w = np.polynomial.legendre.leggauss(ns)
            for _ in range(2):
                jq.eval_f_g_grad(pcof, p, wa, x * 0.05, w * 0.5, True, shift=shift)
            t = wa.last_timing()
            msg += "  split %d: fam %d var %2d bwd %7.2f ms" % (split, t["kernel_family"], t["kernel_variant"], t["ms_backward"])
            wa.close()
        print(msg, flush=True)
