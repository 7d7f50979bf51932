"""Per-family cost of the MFMA register form: the same timing table for the library named by JQ_LIB (default: the shipped one).
Run once per build (VGPR form / default form) and compare: `python scripts/time_forms.py > a.txt; JQ_LIB=.../libjuqbox_hip_df.so python
scripts/time_forms.py > b.txt`.  Every row is the propagator time (HIP events, ms) of the second of two evaluations, shortened problems
(2 000 steps) so that the whole table takes about a minute.  Works with the environment-variable knobs of ABI <= 4 and with the options of
ABI 5 (whichever the loaded library has)."""
import json, os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import juqbox_jl_amd as jq
from test_gpu_random import random_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEW = hasattr(jq._lib.load(), "jq_set_option") if hasattr(jq, "_lib") else False


def make(cls, p, n, env):
    if NEW:
        return cls(p, n, options={k[3:].lower(): int(v) for k, v in env.items()})
    os.environ.update(env)
    try:
        return cls(p, n)
    finally:
        pass


def row(name, p, pcof, ns, env=None, imr=False, ens=None):
    env = env or {}
    if imr:
        p.Integrator_id = jq.Implicit_Midpoint
        p.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=p.N)
    wa = make(jq.Working_Arrays_M_HIP if imr else jq.Working_Arrays_HIP, p, pcof.size, env)
    try:
        if ens is None:
            nodes, weights, shift = (np.zeros(1), np.ones(1), None) if ns == 1 else (0.01 * np.linspace(-1, 1, ns), np.full(ns, 1.0 / ns), np.linspace(0, 1, p.Ntot))
        else:
            nodes, weights, shift = ens
        for _ in range(2):
            jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=shift)
        t = wa.last_timing()
        var = t.get("kernel_variant", t.get("reserved"))
        print("%-44s fam %d <%2d,%2d> var %2d  fwd %9.3f  bwd %9.3f ms" % (name, t["kernel_family"], t["kernel_size"], t["kernel_band"], var, t["ms_forward"], t["ms_backward"]), flush=True)
    finally:
        wa.close()
        for k in env:
            os.environ.pop(k, None)


def cnot3(nsteps=2000):
    p, _ = jq.cases.cnot3()
    p.T, p.nsteps = p.T * nsteps / p.nsteps, nsteps
    pcof = np.array(json.load(open(os.path.join(ROOT, "tests/golden/cnot3.json")))["pcof0"])
    return p, pcof


print(jq._lib.load().jq_version().decode())
for ns, tag in ((1, "u_: cq3 three workgroups"), (100, "u_: cq3 two workgroups"), (200, "u_: cq one workgroup"), (512, "u_ fwd2 + p_: qsplit qw2"),
                (1024, "s_ + p_: qsplit qw4"), (2048, "s_: two slabs / wg"), (3072, "k_6_7: three slabs / wg")):
    p, pcof = cnot3()
    row("cnot3 x %4d  %s" % (ns, tag), p, pcof, ns, ens=jq.cases.cnot3_ensemble(ns) if ns > 1 else None)
p, pcof = cnot3()
row("cnot3 x 4096  k_6_8 slab T4", p, pcof, 4096, {"JQ_QUAD": "0", "JQ_COOP_MAX": "0", "JQ_CQ": "0"}, ens=jq.cases.cnot3_ensemble(4096))
p, pcof = cnot3()
row("cnot3 x 4096  k_6_9 slab OD", p, pcof, 4096, {"JQ_T4": "0", "JQ_COOP_MAX": "0"}, ens=jq.cases.cnot3_ensemble(4096))
p, pcof = cnot3()
row("cnot3 x  256  c_6_9 coop OD", p, pcof, 256, {"JQ_T4": "0"}, ens=jq.cases.cnot3_ensemble(256))
p, pcof = cnot3()
row("cnot3 x 4096  k_6_1 slab band", p, pcof, 4096, {"JQ_OD": "0", "JQ_COOP_MAX": "0"}, ens=jq.cases.cnot3_ensemble(4096))
p, pcof = cnot3()
row("cnot3 x  256  c_6_1 coop band", p, pcof, 256, {"JQ_OD": "0"}, ens=jq.cases.cnot3_ensemble(256))
pd, _ = jq.cases.cnot3_dense()
pd.T, pd.nsteps = pd.T * 2000 / pd.nsteps, 2000
row("cnot3-dense x 4096  k_6_5 slab dense", pd, pcof, 4096, ens=jq.cases.cnot3_ensemble(4096))
p, pcof = cnot3()
row("cnot3 IMR x    1  v_: cq imr3", p, pcof, 1, imr=True)
p, pcof = cnot3()
row("cnot3 IMR x  256  v_: cq imr2", p, pcof, 256, imr=True, ens=jq.cases.cnot3_ensemble(256))
p, pcof = cnot3()
row("cnot3 IMR x 1024  q_: quad imr", p, pcof, 1024, imr=True, ens=jq.cases.cnot3_ensemble(1024))
p, pcof = cnot3()
row("cnot3 IMR x  256  i_6_9 coop imr", p, pcof, 256, {"JQ_T4": "0"}, imr=True, ens=jq.cases.cnot3_ensemble(256))
rng = np.random.default_rng(5)
for Ntot, N, st, ns, env, tag in ((64, 4, True, 4096, {"JQ_COOP_MAX": "0"}, "k_4_1 slab band"), (64, 4, True, 200, {}, "c_4_1 coop band"),
                                  (48, 16, False, 1024, {"JQ_COOP_MAX": "0"}, "k_3_2 slab dense"), (48, 16, False, 200, {}, "c_3_2 coop dense"),
                                  (200, 8, False, 64, {}, "c_13_15 big dense"), (150, 8, True, 64, {}, "c_10_1 big band"),
                                  (112, 4, "t4", 1024, {}, "s_/p_ 7_7 quad"), (32, 4, "t4", 2048, {}, "k_2_7/8")):
    p, pcof = random_problem(jq, rng, Ntot, N, 2, 2, 1000, 4, 1, st)
    row("random Ntot %3d N %2d x %4d  %s" % (Ntot, N, ns, tag), p, pcof, ns, env)
