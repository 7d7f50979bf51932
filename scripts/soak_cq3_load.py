"""Soak of the three-workgroup latency kernels UNDER LOAD: an idle-GPU soak (every result compared bit for bit with the
first one of its kind) while OTHER PROCESSES keep the GPU busy -- one runs the 3 072-sample throughput kernels back to back (every CU
streaming, three waves per SIMD), one runs its own <= 80-sample split grids (a second set of workgroups that wait for each other).
Inside a process the library never lets a split grid share the device (DevGate); across processes nothing can be checked, so this
is where co-residency can really break: a quad whose workgroups are not resident together times out, the evaluation falls back to the
one-workgroup kernel, and the result must STILL be bit-identical.  The consumer side is L1-warm by construction (every role re-reads
the same ring slots every eight steps).
python scripts/soak_cq3_load.py [rounds] [nsteps]   (spawns its two load processes itself; role via JQ_SOAK_ROLE)"""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juqbox_jl_amd as jq
role = os.environ.get("JQ_SOAK_ROLE", "main")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
pcof = np.array(json.load(open(os.path.join(ROOT, "tests/golden/cnot3.json")))["pcof0"])
params, info = jq.cases.cnot3()
if nsteps:      # (0: the reference's full length)
    params.T, params.nsteps = params.T * nsteps / params.nsteps, nsteps
stop = os.path.join(ROOT, "gpurun_out", "soak_stop")
if role == "throughput":      # 3 072 perturbed samples, back to back, until the main process says stop
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    nodes, weights, shift = jq.cases.cnot3_ensemble(3072)
    n = 0
    while not os.path.exists(stop):
        jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
        n += 1
    print("throughput load: %d evaluations of 3 072 samples" % n, flush=True)
    sys.exit(0)
if role == "split":           # its own split grids (80 samples = 240 workgroups), back to back
    wa = jq.Working_Arrays_HIP(params, pcof.size)
    nodes, weights, shift = jq.cases.cnot3_ensemble(80)
    n = fb = 0
    first = None
    while not os.path.exists(stop):
        jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)
        cur = (params.last_infidelity, params.last_leak, params.last_infidelity_grad.tobytes())
        first = first or cur
        assert cur == first, "MISMATCH in the split load process"
        fb += wa.last_timing()["kernel_variant"] != 3
        n += 1
    print("split load: %d evaluations of 80 samples, %d on the one-workgroup kernel, plan: %s" % (n, fb, wa.plan_info()["latency_split"]), flush=True)
    sys.exit(0)
# ---- main: reference results on an idle GPU first, then the soak next to the load processes
if os.path.exists(stop):
    os.remove(stop)
first = {}
was = {}
SIZES = {False: (1, 9, 80, 100), True: (1, 9, 80)}      # (100 samples: TWO workgroups per quad, Stormer-Verlet only -- added in round 5 after the
#                                                          late-start hook had found a slot-reuse race of that kernel which this soak, without the size, could not see)
for imr in (False, True):
    p2, _ = jq.cases.cnot3()
    p2.T, p2.nsteps = params.T, params.nsteps
    if imr:
        p2.Integrator_id = jq.Implicit_Midpoint
        p2.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=p2.N)
    elif os.environ.get("JQ_SOAK_WEIGHTS"):
        # (round 5) the Stormer-Verlet evaluations of the MAIN process carry full leakage weights -- two complex forbidden states: four
        # slots on the split kernels, two more arrays in the ring, the state role's partial dots in LDS.  A fall-back of those runs on the
        # quad-layout kernels (a complex W cannot take the one-workgroup kernel): equal to 1e-12 then, not bit-wise
        rng_w = np.random.default_rng(7)
        fs = rng_w.standard_normal((p2.Ntot, 2)) + 1j * rng_w.standard_normal((p2.Ntot, 2))
        fs /= np.linalg.norm(fs, axis=0)
        Wc = sum((0.5 + 0.5 * k) * np.outer(fs[:, k], np.conj(fs[:, k])) for k in range(2))
        p2.wmat_real, p2.wmat_imag = np.asfortranarray(Wc.real.copy()), np.asfortranarray(Wc.imag.copy())
    wa = (jq.Working_Arrays_M_HIP if imr else jq.Working_Arrays_HIP)(p2, pcof.size)
    was[imr] = (p2, wa)
    for ns in SIZES[imr]:
        nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
        jq.eval_f_g_grad(pcof, p2, wa, nodes, weights, True, shift=shift)
        assert wa.last_timing()["kernel_variant"] == (2 if ns > 80 else 3)
        first[imr, ns] = (p2.last_infidelity, p2.last_leak, p2.last_infidelity_grad.tobytes())
loads = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(rounds), str(nsteps)], env=dict(os.environ, JQ_SOAK_ROLE=r),
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("throughput", "split"))]
time.sleep(25)      # (the load processes import, build their handles and reach steady state)
bad = nfb = nev = 0
t0 = time.time()
worst = 0.0
for r in range(rounds):
    for imr in (False, True):
        p2, wa = was[imr]
        for ns in SIZES[imr]:
            nodes, weights, shift = jq.cases.cnot3_ensemble(ns)
            t1 = time.time()
            jq.eval_f_g_grad(pcof, p2, wa, nodes, weights, True, shift=shift)
            worst = max(worst, time.time() - t1)
            nev += 1
            nfb += wa.last_timing()["kernel_variant"] not in (2, 3)
            cur = (p2.last_infidelity, p2.last_leak, p2.last_infidelity_grad.tobytes())
            same = cur == first[imr, ns]
            if not same and wa.last_timing()["kernel_family"] == 6:      # (weighted run that fell back to the quad-layout kernels)
                f0 = first[imr, ns]
                g0, g1 = np.frombuffer(f0[2]), np.frombuffer(cur[2])
                same = abs(cur[0] - f0[0]) <= 1e-12 * abs(f0[0]) and abs(cur[1] - f0[1]) <= 1e-12 * abs(f0[1]) and np.linalg.norm(g1 - g0) <= 1e-11 * np.linalg.norm(g0)
            if not same:
                bad += 1
                print("MISMATCH imr=%s round %d ns %d" % (imr, r, ns), flush=True)
dt = time.time() - t0
open(stop, "w").close()
for p in loads:
    out, _ = p.communicate(timeout=600)
    print(out.strip().splitlines()[-1] if out.strip() else "(load process: no output)", "| rc", p.returncode)
os.remove(stop)
for imr in (False, True):
    print("plan (%s): %s" % ("implicit midpoint" if imr else "Stormer-Verlet", was[imr][1].plan_info()["latency_split"]))
print("%s%d evaluations (%d rounds x 4 + 3 ensemble sizes (Stormer-Verlet + implicit midpoint) x %d steps) next to the load processes: %d mismatches, %d evaluations on the "
      "one-workgroup kernel (fallback / cooling down), %.0f s, slowest evaluation %.2f s" % ("FULL WEIGHTS (complex, rank 2) in the Stormer-Verlet evaluations: " if os.environ.get("JQ_SOAK_WEIGHTS") else "", nev, rounds, params.nsteps, bad, nfb, dt, worst))
