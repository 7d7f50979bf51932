"""Bit-wise comparison of two builds of the library (JQ_LIB) on the latency-path kernels: cnot3 shortened, several ensemble sizes,
Stormer-Verlet and implicit midpoint, even / odd Neumann counts.  python scripts/cmp_libs.py <other lib.so>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import juqbox_jl_amd as jq
    out = {}
    for m in (6, 5):
        params, info = jq.cases.cnot3()
        params.T, params.nsteps = params.T * 1501 / params.nsteps, 1501
        params.linear_solver.max_iter = m
        pcof = np.array(json.load(open(os.path.join(ROOT, "tests/golden/cnot3.json")))["pcof0"])
        for chunk in ("400", None):
            if chunk:
                os.environ["JQ_CHUNK_STEPS"] = chunk
            wa = jq.Working_Arrays_HIP(params, pcof.size)
            os.environ.pop("JQ_CHUNK_STEPS", None)
            for ns in (1, 5, 80, 81, 256, 300, 512):
                nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
                jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
                out["m%d c%s n%d" % (m, chunk, ns)] = [repr(params.last_infidelity), repr(params.last_leak), [repr(float(x)) for x in params.last_infidelity_grad]]
            f, g, *_ = jq.traceobjgrad(pcof, params, wa, False, True)
            out["m%d c%s single" % (m, chunk)] = [repr(f), [repr(float(x)) for x in g]]
            _, hist, _ = jq.traceobjgrad(pcof, params, wa, True, False)
            out["m%d c%s hist" % (m, chunk)] = repr(float(np.sum(np.abs(hist) * np.arange(hist.size).reshape(hist.shape))))
            wa.close()
    json.dump(out, open(sys.argv[2], "w"))
    sys.exit(0)
res = []
for i, lib in enumerate((None, sys.argv[1])):
    env = dict(os.environ)
    if lib:
        env["JQ_LIB"] = os.path.abspath(lib)
    path = "/tmp/cmp_libs_%d.json" % i
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], env=env, check=True)
    res.append(json.load(open(path)))
bad = [k for k in res[0] if res[0][k] != res[1][k]]
print("%d cases compared, %d differ%s" % (len(res[0]), len(bad), (": " + ", ".join(bad[:8])) if bad else ""))
