"""Kernel experiments on the GPU box: time one cnot3 ensemble evaluation with every library variant built by
scripts/exp_variants.sh and compare the results with the first one.  usage: exp_run.py [nsamples] [tag ...]"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ns = sys.argv[1] if len(sys.argv) > 1 else "3072"
tags = sys.argv[2:] or sorted(os.path.basename(p)[6:-3] for p in glob.glob(os.path.join(ROOT, "juqbox.jl_amd/exp/libjq_*.so")))
code = r'''
import json, os, sys, numpy as np
sys.path.insert(0, %r)
import juqbox_jl_amd as jq
ns = int(%r)
params, info = jq.cases.cnot3()
pcof = np.array(json.load(open(os.path.join(%r, "tests/golden/cnot3.json")))["pcof0"])
nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
wa = jq.Working_Arrays_HIP(params, pcof.size)
best = None
for rep in range(2):
    jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
    t = wa.last_timing()
    if best is None or t["ms_total"] < best["ms_total"]: best = t
print("RES " + json.dumps(dict(ms=best["ms_total"], fwd=best["ms_forward"], bwd=best["ms_backward"], inf=params.last_infidelity,
      leak=params.last_leak, g=float(np.linalg.norm(params.last_infidelity_grad)), g0=params.last_infidelity_grad[:3].tolist())))
''' % (ROOT, ns, ROOT)
ref = None
for tag in tags:
    env = dict(os.environ, JQ_LIB=os.path.join(ROOT, "juqbox.jl_amd/exp/libjq_%s.so" % tag))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    ln = [l for l in r.stdout.splitlines() if l.startswith("RES ")]
    if not ln:
        print("%-14s FAILED: %s" % (tag, r.stderr[-300:]))
        continue
    j = json.loads(ln[-1][4:])
    if ref is None:
        ref = j
    dg = abs(j["g"] - ref["g"]) / ref["g"]
    print("%-14s total %7.1f ms  fwd %6.1f  bwd %6.1f  -> %7.1f evals/s   d_inf %.1e d_grad %.1e" % (
        tag, j["ms"], j["fwd"], j["bwd"], int(ns) / j["ms"] * 1e3, abs(j["inf"] - ref["inf"]), dg), flush=True)
