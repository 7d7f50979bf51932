#!/usr/bin/env python3
"""bench.py -- traceobjgrad evals/sec (forward + discrete adjoint) at the cnot3 Hilbert dimension.

One "step" = one eval_f_g_grad pass of the hot path over the job's batch of ensemble samples (every sample is one
full traceobjgrad evaluation of test/cases/cnot3-setup.jl: Ntot=96, N=4, 32 386 Stormer-Verlet steps, 6 Neumann
terms, golden pcof).  Inputs (operators, pcof, ensemble nodes) are resident in HBM when the timed region starts.

Launch modes
  python bench.py --gpus 1                       one process, one GPU
  python bench.py --gpus N        (N > 1)        this process starts N ranks FIRST (torch.distributed.run, one process
                                                 per GPU, RCCL) -- before it touches any GPU -- and relays their line
  python -m torch.distributed.run ... bench.py --gpus N     the driver's form: RANK/WORLD_SIZE come from the environment
                                                 and must agree with --gpus
  python bench.py --gpus N --single-process      ONE process drives N GPUs through a multi-device library handle
                                                 (jq_create_multi: RCCL all-reduce inside libjuqbox_hip.so)
`value` is weak scaling (--samples-per-gpu samples on every GPU, ONE all-reduce of the packed sums per step); the
strong-scaling figure on a fixed 24 576-sample ensemble is reported beside it (`strong_scaling`).

Prints ONE JSON line on rank 0 (see the contract in the repository brief)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak: 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz
                               # (v_mfma_f64_16x16x4_f64 issues every 64 clk; measured 75.4 TF, probes/)
STRONG_TOTAL_SAMPLES = 24576   # fixed ensemble of the strong-scaling run = 8 GPUs x one full round (3072 samples) each
STRONG_SMALL_SAMPLES = 4096    # second, SUB-SATURATING strong-scaling point: 8 GPUs get 512 samples each (latency regime)
PMC_FILES = ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json")   # rocprofv3 PMC summaries (profiles/); used only when recorded for THIS build


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    # 3072 samples = 768 slabs of 16 columns = three slabs on each of the 256 CUs: one round of the kernels that are fastest
    # for large ensembles (quad layout, 12 waves per workgroup).  Other sizes run too (reported as `other_batch_sizes`).
    ap.add_argument("--samples-per-gpu", type=int, default=int(os.environ.get("JQ_BENCH_SAMPLES", "3072")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip everything outside the timed region (profiling runs)")
    ap.add_argument("--single-process", action="store_true",
                    help="one process, --gpus devices behind one multi-device handle (RCCL inside the library)")
    ap.add_argument("--strong-samples", type=int, default=STRONG_TOTAL_SAMPLES)
    ap.add_argument("--strong-small-samples", type=int, default=STRONG_SMALL_SAMPLES)
    ap.add_argument("--dense-only", type=int, default=0, metavar="SAMPLES",
                    help="profiling runs: only the dense_operator block at this batch size (one full-length evaluation), printed as the line")
    ap.add_argument("--quick-extras", action="store_true",
                    help="tests: keep the strong-scaling points and a one-repetition CPU baseline, skip the other side measurements")
    return ap.parse_args()


def visible_gpu_count(sysfs_glob="/sys/class/kfd/kfd/topology/nodes/*/properties"):
    """GPUs this process may use, WITHOUT initialising HIP in it (the parent only launches the ranks): the KFD topology in sysfs
    (a node with simd_count > 0 is a GPU), cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES the way the
    runtime would.  Falls back to torch's device count (which does not initialise the GPU on this image) when sysfs is not there."""
    import glob
    n = None
    nodes = glob.glob(sysfs_glob)
    if nodes:
        n = 0
        for f in nodes:
            try:
                props = dict(ln.split()[:2] for ln in open(f).read().splitlines() if len(ln.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                n = None
                break
    if n is None:
        import torch
        return torch.cuda.device_count()          # (honours the *_VISIBLE_DEVICES variables itself)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(ids))
    return n


def rank_launch_command(args, argv, port):
    """the launcher command line of an N-rank job (one process per GPU, rendezvous on 127.0.0.1); JQ_BENCH_LAUNCHER replaces
    `python -m torch.distributed.run` (tests: a stub that prints its arguments)"""
    launcher = os.environ.get("JQ_BENCH_LAUNCHER")
    head = launcher.split() if launcher else [sys.executable, "-m", "torch.distributed.run"]
    return head + ["--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
                   os.path.abspath(__file__)] + list(argv)


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start N ranks (one process per GPU) before this process touches a GPU, relay
    the children's output and return their exit code."""
    ndev = visible_gpu_count()
    if ndev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible -- refusing to report a %d-GPU "
                         "number from fewer devices\n" % (args.gpus, ndev, args.gpus))
        return 2
    port = 29500 + os.getpid() % 2000
    cmd = rank_launch_command(args, sys.argv[1:], port)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(params, pcof, nrep=2, all_cores=True):
    """The oracle (C restatement of the reference's sparse Stormer-Verlet path, golden-validated) timed on this box's
    host cores: one core like the reference's serial loop, and all cores over independent samples."""
    from oracle.oracle import Oracle
    orc = Oracle(params)                      # sparse products like the reference's use_sparse=true
    t1 = time.perf_counter()
    for _ in range(nrep):
        orc.traceobjgrad(pcof)
    tc = (time.perf_counter() - t1) / nrep
    out = {"value": 1.0 / tc, "unit": "evals/s", "cores": 1, "kind": "port",
           "sample": "%d x one cnot3 traceobjgrad (1 sample = 4 columns x 32386 steps), C restatement of the reference's "
                     "sparse Stormer-Verlet path, single thread like the reference" % nrep,
           "seconds_per_eval": tc, "host_cores_available": os.cpu_count()}
    # all usable cores: one independent evaluation per worker process (fresh processes: nothing here forks a process that
    # has initialised the GPU).  The script grows the worker count (4, 16, 64, ...) only while it still pays and bounds
    # every level in time -- container affinity masks overstate the cores that can really run.
    if not all_cores:
        return out
    proc = None
    try:
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_ensemble.py"), "--seconds-per-eval", "%.3f" % tc],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        so, _ = proc.communicate(timeout=120)
        j = json.loads(so.strip().splitlines()[-1])
        out["all_cores"] = {"value": j["evals_per_s"], "unit": "evals/s", "cores": j["procs"], "cpu_quota": j.get("cpu_quota"),
                            "sample": "%d concurrent processes x one cnot3 traceobjgrad each (independent ensemble samples)"
                                      % j["procs"], "seconds": j["seconds"]}
    except Exception as e:  # noqa: BLE001 -- a reported baseline must not take the bench down
        if proc is not None and proc.poll() is None:
            import signal
            os.killpg(proc.pid, signal.SIGKILL)        # the script's own process group (workers included), nothing else
        out["all_cores"] = {"error": str(e)[:200]}
    return out


def issue_bound(launch_s, steps_per_launch, products_per_step, waves_per_simd):
    """What the quad-layout formulation itself allows: probes/t4q_issue_probe times the bare product -- per
    wave 6 v_mfma_f64_4x4x4, 22 v_fma_f64, 24 v_mov_b32_dpp, operators in registers, nothing else -- at 1 .. 4 waves per SIMD.  The
    kernel's cycles per product and wave (its whole time step: products, dot products, reductions, updates, staging) over the
    probe's is the fraction of the issue-bound rate it reaches.  Both are converted with the same nominal clock, so the ratio does
    not depend on it.  The probe runs live when its binary is in the tree (build()), else the figures recorded in profiles/ serve."""
    import re
    import subprocess
    txt, src = None, None
    exe = os.path.join(ROOT, "probes", "t4q_issue_probe")
    if os.access(exe, os.X_OK):
        try:
            txt = subprocess.run([exe], capture_output=True, text=True, timeout=60).stdout
            src = "probes/t4q_issue_probe, run by this bench"
        except Exception:  # noqa: BLE001
            txt = None
    if not txt or "mode 15" not in txt:
        try:
            txt = open(os.path.join(ROOT, "profiles", "r03_issue_probes.txt")).read()
            src = "profiles/r03_issue_probes.txt (recorded run of probes/t4q_issue_probe)"
        except OSError:
            return None
    # the probe with EIGHT products per loop iteration (round 3: a loop branch per product cost 30 cycles of the SIMD's time and was
    # in every earlier figure); an older probe output only has the one-product loop
    mm = re.search(r"\[8 products per loop iteration\] mode 15 .*? %d wave\(s\)/SIMD:.*?per wave\s+(\d+) clk" % waves_per_simd, txt)
    unrolled = mm is not None
    if not mm:
        mm = re.search(r"mode 15 .*? %d wave\(s\)/SIMD:.*?per wave\s+(\d+) clk" % waves_per_simd, txt)
    if not mm:
        return None
    probe_clk = float(mm.group(1))
    kernel_clk = launch_s / steps_per_launch / products_per_step * 2.4e9 / waves_per_simd
    return {"bare_product_clk_per_wave": probe_clk, "kernel_clk_per_product_and_wave": kernel_clk,
            "frac_of_formulation_bound": probe_clk / kernel_clk, "waves_per_simd": waves_per_simd,
            "products_per_backward_step": products_per_step, "nominal_issue_model_clk": 280, "source": src,
            "probe_loop": "8 products per loop iteration" if unrolled else "1 product per loop iteration (includes a taken branch)",
            "note": "bare product: 6 MFMA (16 clk) + 24 v_mov_b32_dpp (4.2) + 22 fp64 FMA (4.65, probes/lone_wave_probe.hip) = 299 clk "
                    "nominal; every instruction of the kernel -- scalar, branch, wait -- costs the SIMD ~4.6 clk (DESIGN.md section 6)"}


def dense_operator_block(jq, L, pcof, quick=False, samples=None):
    """north_star's "dense (H x state-batch) contraction": cnot3's dimensions with a dense Hermitian drift (cases.cnot3_dense) -- the
    path every user Hamiltonian without Kronecker structure takes: dense 16 x 16 x 4 fp64 MFMA tiles, k_forward / k_backward<6, 5>, one
    wave per 16-column slab.  Outside the timed region of `value`.  Batch size: the best of a few candidates at 2 000 steps, then ONE
    evaluation at the reference's full length (32 386 steps).  The fraction is EXECUTED fp64 MFMA FLOP (library count, checked against
    the PMC record of profiles/ when it belongs to this build) over HIP-event time over the 78.6 TFLOP/s matrix peak."""
    import numpy as np
    pd, _ = jq.cases.cnot3_dense()
    nfull = pd.nsteps
    cand = [samples] if samples else ([1024] if quick else [1024, 2048, 4096, 8192])
    best, sweep = cand[0], {}
    if len(cand) > 1:
        ps, _ = jq.cases.cnot3_dense()
        ps.T, ps.nsteps = ps.T * 2000 / ps.nsteps, 2000
        ws = jq.Working_Arrays_HIP(ps, pcof.size)
        for ns in cand:
            n2, w2, s2 = jq.cases.cnot3_ensemble(ns)
            jq.eval_f_g_grad(pcof, ps, ws, n2, w2, True, shift=s2)
            jq.eval_f_g_grad(pcof, ps, ws, n2, w2, True, shift=s2)
            t = ws.last_timing()
            sweep[str(ns)] = {"ms_per_2000_steps": t["ms_total"], "evals_per_s_at_full_length": ns / (t["ms_total"] * 1e-3 * nfull / 2000.0),
                              "kernel_family": t["kernel_family"]}
        ws.close()
        top = max(sweep[str(ns)]["evals_per_s_at_full_length"] for ns in cand)
        best = min(ns for ns in cand if sweep[str(ns)]["evals_per_s_at_full_length"] >= 0.995 * top)      # (the smallest batch at the best rate)
    if quick:
        pd.T, pd.nsteps = pd.T * 500 / pd.nsteps, 500
    wd = jq.Working_Arrays_HIP(pd, pcof.size)
    plan = wd.plan_info()
    n2, w2, s2 = jq.cases.cnot3_ensemble(best)
    t1 = time.perf_counter()
    jq.eval_f_g_grad(pcof, pd, wd, n2, w2, True, shift=s2)
    wall = time.perf_counter() - t1
    t = wd.last_timing()
    wd.close()
    kb = "k_backward<%d, %d>" % (t["kernel_size"], t["kernel_band"])
    nb, nf = max(t["n_backward_launches"], 1), max(t["n_forward_launches"], 1)
    bwd_s, fwd_s = t["ms_backward"] * 1e-3 / nb, t["ms_forward"] * 1e-3 / nf
    mf_b, mf_f = t["mfma_backward"] / nb, (t["mfma_executed"] - t["mfma_backward"]) / nf
    libver = L.jq_version().decode()
    src, check, pk = "analytic (library count)", None, {}
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_dense.json")))
        pk = pj["kernels"].get(kb, {})
        if pj.get("library_version") == libver and pk.get("samples_per_gpu") == best and pk.get("mfma_16x16x4_equiv_per_launch") \
                and pk.get("steps_per_launch") in (None, pd.nsteps / nb):
            dev = abs(pk["mfma_16x16x4_equiv_per_launch"] - mf_b) / mf_b
            check = {"pmc": pk["mfma_16x16x4_equiv_per_launch"], "analytic": mf_b, "rel_diff": dev}
            if dev < 0.01:
                mf_b, src = pk["mfma_16x16x4_equiv_per_launch"], "rocprofv3 PMC SQ_INSTS_MFMA (profiles/r06_pmc_dense.json, same build)"
        else:
            pk = {}
    except Exception:  # noqa: BLE001
        pk = {}
    ach_b = mf_b * 2048.0 / bwd_s / 1e12
    ach_f = mf_f * 2048.0 / fwd_s / 1e12
    Ntot, N, Nc, m = pd.Ntot, pd.N, pd.Ncoupled, pd.linear_solver.max_iter
    return {"workload": "cnot3 dimensions (Ntot=96, N=4, %d steps, %d Neumann terms, 3 controls) with a DENSE Hermitian drift "
                        "(cases.cnot3_dense) x %d perturbed samples" % (pd.nsteps, m, best),
            "structure": plan["structure"], "block_band": plan["block_band"], "samples": best, "seconds": wall,
            "evals_per_s": best / (t["ms_total"] * 1e-3), "ms_total": t["ms_total"], "ms_forward": t["ms_forward"], "ms_backward": t["ms_backward"],
            "kernel_family": t["kernel_family"], "kernel": kb + " / k_forward<%d, %d> (dense 16x16x4 fp64 MFMA tiles, one wave per 16-column slab)" % (t["kernel_size"], t["kernel_band"]),
            "batch_size_sweep": sweep,
            "roofline": {"bound": "mfma", "kernel": kb, "achieved": ach_b, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_b / FP64_MFMA_PEAK_TFLOPS, "avg_launch_ms": bwd_s * 1e3, "launches": int(t["n_backward_launches"]),
                         "mfma_count_source": src, "mfma_count_check": check, "valu_per_mfma": pk.get("valu_per_mfma"),
                         "wait_frac": pk.get("wait_frac"), "traffic": pk.get("hbm_bytes_per_launch"),
                         "issue_bound": "one v_mfma_f64_16x16x4 occupies its SIMD's DP pipe for 64 clk and a wave's other VALU work does not "
                                        "overlap it (probes/mfma_valu_overlap_probe): the fraction IS the pipe's MFMA share; 1 - frac is "
                                        "vector updates, LDS waits and the per-operator barrier",
                         "dense_contraction_tflops": 2.0 * Ntot * Ntot * (2 * (9 + 2 * m) + 7 * Nc) * best * N * pd.nsteps / nb / bwd_s / 1e12,
                         "dense_contraction_note": "SURVEY.md 8(d)'s count (2 Ntot^2 per product and column, 9 + 2m products per step as the "
                                                   "reference forms them); the kernels merge products (8 + 2m) and the trace operators keep their "
                                                   "own band structure, so fewer tiles are executed than this count implies"},
            "roofline_forward": {"kernel": kb.replace("backward", "forward"), "achieved": ach_f, "frac": ach_f / FP64_MFMA_PEAK_TFLOPS,
                                 "avg_launch_ms": fwd_s * 1e3},
            "all_propagators_mfma_frac": t["mfma_executed"] * 2048.0 / (t["ms_propagate"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS}


def lone_wave_chain_clk():
    """probes/lone_wave_probe: clk per link of a DEPENDENT v_fmac_f64 chain of a wave that is alone on its SIMD (one chain / two
    chains interleaved, unroll 16) -- what bounds the row-lane kernels, whose product is two interleaved accumulator chains of
    NPJ / 2 links.  Run live when the binary is in the tree (build()), else the recorded run of profiles/r03_issue_probes.txt."""
    import re
    txt, src = None, None
    exe = os.path.join(ROOT, "probes", "lone_wave_probe")
    if os.access(exe, os.X_OK):
        try:
            txt = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
            src = "probes/lone_wave_probe, run by this bench"
        except Exception:  # noqa: BLE001
            txt = None
    if not txt or "two chains interleaved" not in txt:
        try:
            txt = open(os.path.join(ROOT, "profiles", "r03_issue_probes.txt")).read()
            src = "profiles/r03_issue_probes.txt (recorded run of probes/lone_wave_probe)"
        except OSError:
            return None
    out = {"source": src}
    for key, label in (("one_chain", "ONE dependent chain"), ("two_chains", "two chains interleaved")):
        mm = re.search(r"v_fmac_f64, %s\s+unroll 16, 1 wave\(s\)/SIMD:\s+([0-9.]+) clk" % label, txt)
        if not mm:
            return None
        out[key] = float(mm.group(1))
    return out


def baseline_configs_block(jq, L, quick=False):
    """BASELINE.json configs[0], [1], [2], [4] (rabi, cnot1, cnot2, SWAP-02 risk-neutral x 512 nodes; configs[3] = cnot3 is the bench
    line itself): per config the GPU time of one evaluation (HIP events), the CPU oracle on ONE core in the same run (BASELINE.md
    section 3; a bounded sample of the 512 nodes for the ensemble, stated), and a bound statement.  These Hilbert spaces (Ntot <= 12)
    run on the VALU row-lane kernels: no MFMA percentage is quoted; a single evaluation is the latency of a sequential
    chain of dependent products, so the statement is ACHIEVED clk per dependent product against the dependent-FMA chain latency a lone
    wave can reach (probes/lone_wave_probe): a product of row length NPJ is two interleaved accumulator chains + one add."""
    import copy
    import numpy as np
    from oracle.oracle import Oracle
    chain = lone_wave_chain_clk()
    ncu = 256
    cfgs = {}
    for cname, nq in (("rabi", 1), ("cnot1", 1), ("cnot2", 1), ("swap02_rn", 512)):
        pc, ic = jq.cases.BUILDERS[cname]()
        pcf = ic.get("pcof0")
        if ic.get("golden"):
            gj = json.load(open(os.path.join(ROOT, "tests", "golden", "%s.json" % ic["golden"])))
            pcf = np.array(gj["pcof0"]) if "pcof0" in gj else pcf
        pcf = np.asarray(pcf, dtype=np.float64)
        wc = jq.Working_Arrays_HIP(pc, pcf.size)
        ncu = L.jq_num_compute_units(wc.handle) or ncu
        if nq == 1:
            nd, wq, shc = np.zeros(1), np.ones(1), None
        else:
            nd, wq, shc = ic["nodes"], ic["weights"], pc.shift_weights_reference()
        for _ in range(2):
            jq.eval_f_g_grad(pcf, pc, wc, nd, wq, True, shift=shc)
        tc = wc.last_timing()
        gpu_infid = float(pc.last_infidelity)
        m, nsteps, N = int(pc.linear_solver.max_iter), int(pc.nsteps), int(pc.N)
        entry = {"Ntot": int(pc.Ntot), "N": N, "nsteps": nsteps, "neumann_terms": m, "samples": nq,
                 "ms_per_evaluation": tc["ms_total"], "ms_forward": tc["ms_forward"], "ms_backward": tc["ms_backward"],
                 "evals_per_s": nq / (tc["ms_total"] * 1e-3), "svts_per_s": tc["svts"] / (tc["ms_total"] * 1e-3),
                 "kernel_family": tc["kernel_family"], "kernel_size": tc["kernel_size"]}
        if nq > 1:      # strong scaling of THIS ensemble over 8 GPUs, predicted from one GPU: a rank's share is nq / 8 nodes
            nsh = nq // 8
            for _ in range(2):
                jq.eval_f_g_grad(pcf, pc, wc, nd[:nsh], wq[:nsh], True, shift=shc)
            t8 = wc.last_timing()["ms_total"]
            entry["strong_scaling_prediction_8_gpus"] = {
                "ms_one_gpu_all_nodes": tc["ms_total"], "ms_one_rank_share": t8, "nodes_per_rank": nsh, "predicted_speedup": tc["ms_total"] / t8,
                "note": "the %d-node ensemble does not fill ONE GPU (%d columns on %d compute units): every node runs at the latency of its "
                        "sequential time loop, and an eighth of the nodes takes as long as all of them -- splitting it over 8 GPUs buys "
                        "~ 1 x, not 6 x (SURVEY.md 8(e)); north_star's >= 6 x needs a saturating ensemble (strong_scaling)" % (nq, nq * N, ncu)}
        wc.close()
        # ---- CPU oracle, one core, same run (dense products: these cases are use_sparse = false in the reference) ----
        ncpu = nq if nq == 1 else (4 if quick else 16)       # bounded sample of the ensemble's nodes, scaled to the whole ensemble
        orc = Oracle(pc)
        reps = 1 if quick else (5 if nq == 1 else 1)
        times, cpu_infid = [], 0.0
        H0 = pc.Hconst.copy()
        for _ in range(reps):
            t1 = time.perf_counter()
            if nq == 1:
                cpu_infid = orc.traceobjgrad(pcf)["primaryobjf"]
            else:
                sub = np.linspace(0, nq - 1, ncpu).round().astype(int)      # nodes spread over the whole interval
                r = orc.eval_f_g_grad(pcf, nd[sub], wq[sub], shc, True)
                cpu_infid = r["last_infidelity"]
            times.append(time.perf_counter() - t1)
        pc.Hconst = H0
        tcpu = float(np.median(times)) * (nq / ncpu)
        entry["cpu_baseline"] = {"value": nq / tcpu, "unit": "evals/s", "cores": 1, "kind": "port", "seconds_per_evaluation": tcpu,
                                 "sample": ("median of %d x one traceobjgrad (C restatement of the reference's Stormer-Verlet path, one thread)" % reps) if nq == 1 else
                                           ("%d of the %d quadrature nodes (spread over the interval), one eval_f_g_grad loop on one thread, time "
                                            "scaled by %d / %d" % (ncpu, nq, nq, ncpu))}
        entry["gpu_over_cpu_1core"] = tcpu / (tc["ms_total"] * 1e-3)
        if nq == 1:      # the same evaluation on both sides: a live parity check of the line's own numbers at the reference's criterion
            d = abs(gpu_infid - cpu_infid)      # (test/evalGrad.jl:43-67: difference below atol 1e-14, or relative difference below rtol 1e-10)
            entry["infidelity"] = {"gpu": gpu_infid, "cpu_oracle": cpu_infid, "abs_diff": d,
                                   "passes_reference_criterion": bool(d < 1e-14 or d / max(abs(cpu_infid), 1e-300) < 1e-10)}
        # ---- bound: latency of the dependent chain --------------------------------------------------------------
        if tc["kernel_family"] == 3 and chain:
            npj = int(tc["kernel_size"])
            nwaves = -(-(nq * N) // 4)
            variant = int(tc["kernel_variant"])                      # 33: state | adjoint | traces on three (four) waves, 32: two waves, 0: one
            prods = 8 + 2 * m                                        # dependent products of one Stormer-Verlet step
            prods_b = {33: 10 + 2 * m, 32: 10 + 2 * m + 4 * int(pc.Ncoupled)}.get(variant, 18 + 4 * m + 4 * int(pc.Ncoupled))   # longest wave of the backward sweep
            clk_f = tc["ms_forward"] * 1e-3 * 2.4e9 / (nsteps * prods)
            clk_b = tc["ms_backward"] * 1e-3 * 2.4e9 / (nsteps * prods_b)
            # two interleaved chains of NPJ / 2 links each (two_chains clk per instruction, NPJ instructions) + the add that joins them
            bound = npj * chain["two_chains"] + chain["one_chain"]
            entry["bound"] = {"kind": "latency of the dependent fp64 FMA chain of a lone wave (no MFMA percentage at Ntot <= 16: SURVEY.md 8(d))",
                              "dependent_products_per_step_forward": prods, "products_per_step_longest_backward_wave": prods_b, "row_length_NPJ": npj,
                              "backward_kernel_variant": variant,
                              "clk_per_dependent_product_forward": clk_f, "clk_per_dependent_product_backward": clk_b,
                              "chain_latency_bound_clk_per_product": bound,
                              "frac_forward": bound / clk_f, "frac_backward": bound / clk_b,
                              "probe": chain,
                              "note": "bound = NPJ x (clk per link of two interleaved dependent v_fmac_f64 chains) + one dependent add; the "
                                      "remainder is the operand reads from the LDS ring, register moves in front of the accumulate-form FMAs, the leak "
                                      "integrand, s_nop hazard slots, and in the backward sweep one workgroup barrier and the LDS hand-off per step"}
        # the same with the implicit-midpoint integrator (the default of the reference's example scripts)
        if cname != "rabi":
            pmc_ = copy.copy(pc)
            pmc_.Integrator_id = jq.Implicit_Midpoint
            pmc_.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=pmc_.N)
            wmc = jq.Working_Arrays_M_HIP(pmc_, pcf.size)
            for _ in range(2):
                jq.eval_f_g_grad(pcf, pmc_, wmc, nd, wq, True, shift=shc)
            entry["ms_per_evaluation_implicit_midpoint"] = wmc.last_timing()["ms_total"]
            wmc.close()
        cfgs[cname] = entry
    cfgs["cnot3"] = "the bench line itself (value / roofline / cpu_baseline; single_evaluation for one sample)"
    return cfgs


_T0 = time.perf_counter()


def _trace(msg):
    if os.environ.get("JQ_BENCH_TRACE"):
        sys.stderr.write("[bench %7.1f s] %s\n" % (time.perf_counter() - _T0, msg))
        sys.stderr.flush()


def main():
    args = parse_args()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched and not args.single_process:
        sys.exit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if launched and world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d does not match WORLD_SIZE=%d of the launcher\n" % (args.gpus, world))
        sys.exit(2)
    if args.single_process and launched and world > 1:
        sys.stderr.write("bench.py: --single-process cannot run under a multi-rank launcher\n")
        sys.exit(2)

    import numpy as np
    import torch
    import juqbox_jl_amd as jq
    from juqbox_jl_amd import _lib
    from juqbox_jl_amd.ipopt_interface import shard_bounds

    _trace("imports done")
    L = _lib.load()
    ndev_visible = L.jq_device_count()
    ngpus = args.gpus                                  # GPUs of the whole job
    # TEST MODE of jq_create_multi (include/juqbox_hip.h): the sub-handles share the visible GPU(s), the all-reduce is a host-side
    # sum -- the line is labelled as such and is NOT a multi-GPU measurement; it exists so that this reporting code runs on a one-GPU box
    same_device = args.single_process and "multi_same_device=1" in os.environ.get("JQ_OPTIONS", "").replace(" ", "")
    if args.single_process:
        if ndev_visible < (1 if same_device else ngpus):
            sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible\n" % (ngpus, ndev_visible))
            sys.exit(2)
    elif ndev_visible <= local_rank:
        sys.stderr.write("bench.py: rank %d has no GPU (%d visible)\n" % (rank, ndev_visible))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    _lib.check(L.jq_set_device(local_rank))
    dist = None
    if launched:                                       # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    params, info = jq.cases.cnot3()
    pcof = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "cnot3.json")))["pcof0"])
    if args.dense_only:
        print(json.dumps({"dense_operator": dense_operator_block(jq, L, pcof, samples=args.dense_only)}), flush=True)
        return
    nsamples_total = args.samples_per_gpu * ngpus
    nodes, weights, shift = jq.cases.cnot3_ensemble(nsamples_total)
    wa = jq.Working_Arrays_HIP(params, pcof.size, devices=ngpus if args.single_process else None)
    _trace("handle created")

    def step():
        jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)

    def sync_all():
        for d in range(min(ngpus, ndev_visible) if args.single_process else 1):
            torch.cuda.synchronize(d if args.single_process else local_rank)

    def fence():
        sync_all()
        if dist is not None:
            dist.barrier()
        sync_all()

    per_rank = [0.0, 0.0]       # ms per step of the fastest / slowest rank in the last timed() region (before its closing barrier)
    ar_ms = []                  # all-reduce wall times of the last timed() region

    def timed(fn, nsteps):
        """barrier + synchronize, nsteps x fn, barrier + synchronize; MAX over ranks of the elapsed time"""
        fence()
        t0 = time.perf_counter()
        acc = []
        ar_ms.clear()
        for _ in range(nsteps):
            fn()
            acc.append(wa.last_timing())
            # the ONE all-reduce of the step: inside the library (multi-device handle) or torch.distributed's (one process per GPU)
            ar_ms.append(acc[-1]["ms_allreduce"] if args.single_process else wa.last_allreduce_ms)
        if args.single_process:     # devices of the handle: slowest / fastest shard of the last step
            per_rank_dev = (acc[-1]["ms_shard_min"], acc[-1]["ms_shard_max"])
        el_rank = time.perf_counter() - t0        # this rank's own compute (before the closing barrier)
        fence()
        el = time.perf_counter() - t0
        el_min = el_max = el_rank
        if dist is not None:
            t = torch.tensor([el, el_rank, -el_rank], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el, el_max, el_min = float(t[0].item()), float(t[1].item()), -float(t[2].item())
        per_rank[:] = [el_min / nsteps * 1e3, el_max / nsteps * 1e3]
        if args.single_process:
            per_rank[:] = list(per_rank_dev)
        return el, acc

    for _ in range(args.warmup):
        step()
    _trace("warm-up done")
    elapsed, tms = timed(step, args.steps)
    per_rank_ms = {"min": per_rank[0], "max": per_rank[1],
                   "note": "ms per step of the fastest / slowest rank (device, for --single-process) before the closing barrier"}
    allreduce_ms = sum(ar_ms) / max(len(ar_ms), 1)
    _trace("timed steps done")
    prop_ms = sum(t["ms_propagate"] for t in tms)
    bwd_ms = sum(t["ms_backward"] for t in tms)
    fwd_ms = sum(t["ms_forward"] for t in tms)
    nb = sum(t["n_backward_launches"] for t in tms)
    mfma = sum(t["mfma_executed"] for t in tms)
    mfma_bwd = sum(t["mfma_backward"] for t in tms)
    tm = tms[-1]
    infid_weak = params.last_infidelity

    # ---- strong scaling: a FIXED ensemble split over the job's GPUs (outside the timed region of `value`) ------------
    strong = strong_small = None
    if not args.no_extras:
        def strong_point(ns, note):
            n2, w2, s2 = jq.cases.cnot3_ensemble(ns)

            def strong_step():
                jq.eval_f_g_grad(pcof, params, wa, n2, w2, True, shift=s2)
            strong_step()                                   # (buffers grow to the shard size here)
            el2, tms2 = timed(strong_step, 1)
            return {"total_samples": ns, "samples_per_gpu": ns // ngpus, "seconds": el2, "evals_per_s": ns / el2,
                    "per_rank_ms": {"min": per_rank[0], "max": per_rank[1]}, "allreduce_ms": sum(ar_ms) / max(len(ar_ms), 1),
                    "kernel_family": tms2[-1]["kernel_family"], "note": note}
        strong = strong_point(args.strong_samples, "fixed ensemble, block-partitioned over the GPUs, one all-reduce; compare "
                              "evals_per_s across n_gpus.  Saturating: every GPU keeps >= one full round (3 072 samples) up to 8 GPUs: "
                              "predicted speed-up at 8 GPUs 8 x (a rank's share is exactly the weak-scaling workload of `value`)")
        _trace("strong-scaling run done")
        # the domain of north_star's '>= 6x at 8 GPUs' made explicit: a fixed ensemble that does NOT saturate 8 GPUs.  One GPU
        # needs 1.5 rounds of the throughput kernels; 8 GPUs run 512 samples each in the latency regime (one time loop's
        # latency), so the speed-up is bounded by (time of 4 096 on one GPU) / (latency of one 512-sample round) ~ 4-5 x
        strong_small = strong_point(args.strong_small_samples, "SUB-SATURATING fixed ensemble: at 8 GPUs each rank holds 512 samples "
                                    "and runs at the latency of one time loop; predicted speed-up at 8 GPUs = this time / the 512-sample "
                                    "time of mid_size_ensembles ~ 4.8 x, not 8 x; the SWAP-02 risk-neutral ensemble of BASELINE config 5 "
                                    "(512 nodes): ~ 1 x (baseline_configs.swap02_rn.strong_scaling_prediction_8_gpus; DESIGN.md section 8)")
        _trace("small strong-scaling run done")

    if rank == 0:
        evals = nsamples_total * args.steps
        value = evals / elapsed
        Ntot, N, Nc, m, nsteps = params.Ntot, params.N, params.Ncoupled, params.linear_solver.max_iter, params.nsteps
        lo, hi = shard_bounds(nsamples_total, 0, ngpus)
        svts_rank = (hi - lo) * N * nsteps * args.steps                      # SURVEY.md section 8(d), one GPU's share
        f_bwd = 2.0 * Ntot * Ntot * (2 * (9 + 2 * m) + 7 * Nc)                 # dense-contraction FLOP / SVTS, backward sweep
        f_fwd = 2.0 * Ntot * Ntot * (9 + 2 * m)
        avg_launch_s = bwd_ms * 1e-3 / max(nb, 1)
        fam = {0: "k_backward", 1: "k_backward_coop", 2: "k_backward_lane", 3: "k_backward_rowlane", 4: "k_backward_rowlane_imr",
               5: "k_backward_coop_imr", 6: "k_backward", 7: "k_backward_quad_imr", 8: "k_backward_cq", 9: "k_backward_cq_imr"}.get(
                   tm.get("kernel_family", 0), "k_backward")
        kname = "%s<%d, %d>" % (fam, tm.get("kernel_size", 0), tm.get("kernel_band", 0))
        band = tm.get("kernel_band")
        band_note = {9: " (band 9 = block tridiagonal with diagonal off-diagonal blocks)",
                     8: " (band 8 = 4x4 diagonal blocks on v_mfma_f64_4x4x4 + diagonal couplings on DPP FMAs)",
                     7: " (band 7 = the band-8 product in the quad layout: four columns per wave, a 16-row block per register; "
                        "%d waves per workgroup)" % (4 * max(1, round(args.samples_per_gpu * N / 16 / 256)) if args.samples_per_gpu * N / 16 <= 768 else 12)}.get(band, "")
        # ---- roofline of the dominant kernel (k_backward): EXECUTED fp64 matrix FLOP / HIP-event time / matrix peak.
        # The kernels skip the structural zeros of the operators, so the executed arithmetic -- not the dense-contraction
        # count of SURVEY.md 8(d), which would exceed the peak -- is what the fraction is made of.  MFMA instruction
        # count: rocprofv3 PMC (SQ_INSTS_MFMA) of this build when profiles/ holds one, else the library's analytic count.
        NT = (Ntot + 15) // 16
        flop_per_mfma = 2048.0                          # jq_timing counts in units of one v_mfma_f64_16x16x4 (4 x v_mfma_f64_4x4x4_4b)
        mfma_src = "analytic (library count)"
        pmc, pmc_file = {}, None
        libver = L.jq_version().decode()                # carries the SHA-256 prefix of the library's sources
        for fn in PMC_FILES:
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", fn)))
                pk = pj["kernels"].get(kname)
                # a PMC record is joined ONLY when it was taken from exactly this build (source hash in jq_version()) and workload
                if pk and pj.get("library_version") == libver and pk.get("samples_per_gpu") == args.samples_per_gpu:
                    pmc, pmc_file = pk, fn
                    break
            except Exception:  # noqa: BLE001
                pass
        mfma_analytic = mfma_bwd / max(nb, 1)
        mfma_bwd_per_launch = mfma_analytic
        mfma_check = None
        steps_launch_now = nsteps * args.steps / max(nb, 1)
        if pmc.get("mfma_16x16x4_equiv_per_launch") and pmc.get("steps_per_launch") in (None, steps_launch_now):
            # (joined only for the same chunking: a record without steps_per_launch is an older one taken at the default chunking)
            mfma_pmc = pmc["mfma_16x16x4_equiv_per_launch"]
            dev = abs(mfma_pmc - mfma_analytic) / mfma_analytic
            mfma_check = {"pmc": mfma_pmc, "analytic": mfma_analytic, "rel_diff": dev, "mismatch": dev >= 0.01}
            if dev < 0.01:
                mfma_bwd_per_launch = mfma_pmc
                mfma_src = "rocprofv3 PMC SQ_INSTS_MFMA (profiles/%s, same build: %s)" % (pmc_file, libver)
            else:      # a side measurement must not take the judged line down: keep the analytic count and say so
                mfma_src = "analytic (library count); the PMC record of profiles/%s disagrees by %.2f %% and is NOT used" % (pmc_file, 100 * dev)
        elif pmc:
            mfma_check = {"pmc": None, "analytic": mfma_analytic, "mismatch": None,
                          "note": "PMC record taken with another chunking (steps per launch %s, now %s): not joined" % (pmc.get("steps_per_launch"), steps_launch_now)}
            pmc = {}
        achieved = mfma_bwd_per_launch * flop_per_mfma / avg_launch_s / 1e12
        traffic_alg = None
        if band in (7, 8):       # algorithmic HBM bytes of one k_backward launch
            steps_launch = steps_launch_now
            nslabs_rank = -(-(hi - lo) * N // 16)
            traffic_alg = ((2 * steps_launch + 1) * 2 * 128 * NT * 8           # tile stream: Kp/Kn and S image per time point
                           + 2 * nslabs_rank * (4 * 4 * NT + 8) * 64 * 8    # state file of every slab (U, V, MU, NU + carries), read and written
                           + steps_launch * 5 * Nc * 8)                         # one trace record per time step
        # coupling FMAs of the products (v_fma_f64 of 64 lanes): 6 NT + 8 (NT - 1) per product and slab in the slab layout,
        # 4 (4 NT - 2) in the quad layout; one product per 4 NT of the 512-FLOP MFMAs
        fma_per_product = (6 * NT + 8 * (NT - 1)) if band == 8 else 4 * (4 * NT - 2) if band == 7 else 0
        executed_fma_bwd = (mfma_bwd_per_launch * 4.0 / (4 * NT)) * fma_per_product * 128.0
        frac = achieved / FP64_MFMA_PEAK_TFLOPS
        assert frac <= 1.0, "roofline fraction above 1: the MFMA count or the peak is wrong"
        roofline = {"bound": "mfma", "kernel": kname + band_note,
                    "achieved_definition": "EXECUTED fp64 MFMA FLOP of one k_backward launch (v_mfma_f64_4x4x4_4b x 512 FLOP; "
                                           "structural zeros are skipped, not counted) / average HIP-event duration of the launch",
                    "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": frac,
                    "mfma_count_source": mfma_src, "mfma_count_check": mfma_check, "library_version": libver,
                    "traffic": pmc.get("hbm_bytes_per_launch"), "traffic_unit": "HBM bytes per launch (PMC)",
                    "traffic_algorithmic": traffic_alg,
                    "traffic_ratio": (pmc["hbm_bytes_per_launch"] / traffic_alg) if pmc.get("hbm_bytes_per_launch") and traffic_alg else None,
                    "traffic_algorithmic_note": "per k_backward launch: the K(t)/S(t) tile stream of the chunk read once (2 images x "
                                                "128 NT doubles per time point), the per-slab state file in and out, one trace record "
                                                "(5 Nc doubles) per time step",
                    "valu_per_mfma": pmc.get("valu_per_mfma"), "wait_frac": pmc.get("wait_frac"),
                    "wait_inst_frac": pmc.get("wait_inst_frac"),
                    "launches": int(nb), "avg_launch_ms": avg_launch_s * 1e3,
                    "frac_incl_coupling_fma": (achieved + executed_fma_bwd / avg_launch_s / 1e12) / FP64_MFMA_PEAK_TFLOPS,
                    "dense_equivalent_tflops": f_bwd * svts_rank / max(nb, 1) / avg_launch_s / 1e12,
                    "dense_equivalent_note": "2 Ntot^2 FLOP per product and column (SURVEY.md 8(d)): what a dense formulation "
                                             "would have to execute for the same result; not a hardware utilisation",
                    "all_propagators_mfma_frac": mfma * flop_per_mfma / (prop_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                    "forward_ms": fwd_ms, "backward_ms": bwd_ms}
        # ... and of the forward propagator (27 % of the step; no trace products: a better fraction than the backward kernel's, which
        # the whole-step figure all_propagators_mfma_frac hides)
        nf = sum(t["n_forward_launches"] for t in tms)
        fwd_launch_s = fwd_ms * 1e-3 / max(nf, 1)
        mfma_fwd_launch = (mfma - mfma_bwd) / max(nf, 1)
        fname = kname.replace("k_backward", "k_forward")
        pf, fsrc = {}, "analytic (library count)"
        if pmc_file:
            try:
                pf = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))["kernels"].get(fname, {})
            except Exception:  # noqa: BLE001
                pf = {}
        if pf.get("mfma_16x16x4_equiv_per_launch") and pf.get("steps_per_launch") in (None, nsteps * args.steps / max(nf, 1)):
            devf = abs(pf["mfma_16x16x4_equiv_per_launch"] - mfma_fwd_launch) / max(mfma_fwd_launch, 1.0)
            if devf < 0.01:
                mfma_fwd_launch = pf["mfma_16x16x4_equiv_per_launch"]
                fsrc = "rocprofv3 PMC SQ_INSTS_MFMA (profiles/%s, same build)" % pmc_file
        ach_f = mfma_fwd_launch * flop_per_mfma / fwd_launch_s / 1e12 if fwd_launch_s > 0 else 0.0
        fma_fwd = (mfma_fwd_launch * 4.0 / (4 * NT)) * fma_per_product * 128.0
        roofline_forward = {"bound": "mfma", "kernel": fname, "achieved": ach_f, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach_f / FP64_MFMA_PEAK_TFLOPS, "launches": int(nf), "avg_launch_ms": fwd_launch_s * 1e3,
                            "mfma_count_source": fsrc, "valu_per_mfma": pf.get("valu_per_mfma"), "wait_frac": pf.get("wait_frac"),
                            "traffic": pf.get("hbm_bytes_per_launch"),
                            "frac_incl_coupling_fma": (ach_f + fma_fwd / fwd_launch_s / 1e12) / FP64_MFMA_PEAK_TFLOPS if fwd_launch_s > 0 else None,
                            "products_per_step": 8 + 2 * m}
        if band == 7 and not args.no_extras and not args.quick_extras:
            roofline["issue_bound"] = issue_bound(avg_launch_s, nsteps * args.steps / max(nb, 1), 2 * (8 + 2 * m) + 4 * Nc,
                                                  min(3, max(1, round(args.samples_per_gpu * N / 16 / 256))))
        out = {"metric": "traceobjgrad evals/sec (fwd+adjoint), cnot3 Hilbert dim", "value": value,
               "unit": "evals/s", "n_gpus": ngpus, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "cnot3 (test/cases/cnot3-setup.jl: Ntot=96, N=4, nsteps=32386, 6 Neumann terms, "
                                      "golden pcof) x risk-neutral ensemble of %d samples per GPU" % args.samples_per_gpu,
                          "samples_per_gpu": args.samples_per_gpu, "columns_per_gpu": args.samples_per_gpu * N,
                          "svts_per_step_all_gpus": nsamples_total * N * nsteps, "parallelism": "ensemble-dp%d" % ngpus,
                          "launcher": ("TEST MODE same-device: one process, %d sub-handles of jq_create_multi on %d physical GPU(s), "
                                       "host-side sum in place of the all-reduce (JQ_OPTIONS=multi_same_device=1) -- not a multi-GPU measurement"
                                       % (wa.num_devices, ndev_visible)) if same_device else
                                      ("one process, %d devices behind one jq_create_multi handle (RCCL all-reduce inside the "
                                       "library)" % wa.num_devices) if args.single_process else
                                      ("torch.distributed: %d rank(s), one process per GPU, backend %s"
                                       % (dist.get_world_size(), dist.get_backend()) if dist is not None else "one process, one GPU"),
                          "ranks": world, "devices_behind_handle": wa.num_devices,
                          # what the collective library itself reports: torch.distributed's world size, or ncclCommCount of the
                          # library's own communicator (--single-process; 0 in the same-device test mode, which has none)
                          "rccl_world_size": dist.get_world_size() if dist is not None else
                                      (wa.rccl_world_size if args.single_process else 1)},
               "svts_per_s": nsamples_total * N * nsteps * args.steps / elapsed,
               "ensemble_infidelity": infid_weak,
               "per_rank_ms": per_rank_ms, "allreduce_ms": allreduce_ms,
               "roofline": roofline, "roofline_forward": roofline_forward}
        if strong is not None:
            out["strong_scaling"] = strong
            out["strong_scaling_small"] = strong_small
        if not args.no_extras and not args.no_cpu_baseline and (ngpus > 1 or args.quick_extras):
            # rank 0 of an N-rank job (the line a SCALE run parses) carries the CPU baseline too: bounded sample, one core
            out["cpu_baseline"] = cpu_baseline(params, pcof, nrep=1 if args.quick_extras else 2, all_cores=False)
        if ngpus == 1 and not args.no_extras and not args.quick_extras:
            # outside the timed region: the latency of ONE evaluation and of the reference's 9-node ensemble
            # (examples/Risk_Neutral/run_all.jl:134) -- what an Ipopt iteration of the reference waits for -- next to the
            # CPU figures below; not part of `value`
            jq.traceobjgrad(pcof, params, wa, False, True)
            t1 = time.perf_counter()
            jq.traceobjgrad(pcof, params, wa, False, True)
            torch.cuda.synchronize()
            ts = wa.last_timing()
            out["single_evaluation"] = {"seconds": time.perf_counter() - t1, "ms_propagate": ts["ms_propagate"],
                                        "kernel_family": ts["kernel_family"], "kernel_band": ts["kernel_band"]}
            n9, w9, s9 = jq.cases.cnot3_ensemble(9)
            jq.eval_f_g_grad(pcof, params, wa, n9, w9, True, shift=s9)
            t1 = time.perf_counter()
            jq.eval_f_g_grad(pcof, params, wa, n9, w9, True, shift=s9)
            out["nine_node_ensemble"] = {"seconds": time.perf_counter() - t1, "evals_per_s": 9 / (time.perf_counter() - t1),
                                         "kernel_family": wa.last_timing()["kernel_family"]}
            try:      # ... and of one evaluation with FULL leakage weights (use_custom_forbidden, src/evalobjgrad.jl:214-232): two real
                      # forbidden states (latency path: cooperative-quad kernels with the low-rank terms) / two complex ones (quad layout)
                import copy
                fw = {}
                for tag, cplx in (("real_rank2", False), ("complex_rank2", True)):
                    pw = copy.copy(params)
                    rng_w = np.random.default_rng(12)
                    fs = rng_w.standard_normal((pw.Ntot, 2)) + (1j * rng_w.standard_normal((pw.Ntot, 2)) if cplx else 0)
                    fs = fs / np.linalg.norm(fs, axis=0)
                    W = sum((0.5 + 0.5 * k) * np.outer(fs[:, k], np.conj(fs[:, k])) for k in range(2))
                    pw.wmat_real, pw.wmat_imag = np.asfortranarray(W.real.copy()), np.asfortranarray(W.imag.copy())
                    ww = jq.Working_Arrays_HIP(pw, pcof.size)
                    jq.traceobjgrad(pcof, pw, ww, False, True)
                    t1 = time.perf_counter()
                    jq.traceobjgrad(pcof, pw, ww, False, True)
                    fw[tag] = {"seconds": time.perf_counter() - t1, "kernel_family": ww.last_timing()["kernel_family"]}
                    ww.close()
                out["single_evaluation_full_weights"] = fw
            except Exception as e:  # noqa: BLE001
                out["single_evaluation_full_weights"] = {"error": str(e)[:200]}
            try:      # the reference's other integrator (implicit midpoint, the default of its examples) on the same problem
                import copy
                pm = copy.copy(params)
                pm.Integrator_id = jq.Implicit_Midpoint
                pm.linear_solver = jq.lsolver_object(solver=jq.JACOBI_SOLVER_M, max_iter=100, tol=1e-12, nrhs=pm.N)
                wm = jq.Working_Arrays_M_HIP(pm, pcof.size)
                jq.traceobjgrad(pcof, pm, wm, False, True)
                t1 = time.perf_counter()
                jq.traceobjgrad(pcof, pm, wm, False, True)
                out["single_evaluation_implicit_midpoint"] = {"seconds": time.perf_counter() - t1,
                                                              "kernel_family": wm.last_timing()["kernel_family"],
                                                              "solver": "jacobi_midpoint, max_iter 100, tol 1e-12 (test/runtests.jl:69-70)"}
                # ... and its ensemble throughput on the same 3 072-sample workload (the examples' default integrator)
                jq.eval_f_g_grad(pcof, pm, wm, nodes, weights, True, shift=shift)
                t1 = time.perf_counter()
                jq.eval_f_g_grad(pcof, pm, wm, nodes, weights, True, shift=shift)
                dt_m = time.perf_counter() - t1
                out["ensemble_implicit_midpoint"] = {"samples": int(nodes.size), "seconds": dt_m, "evals_per_s": nodes.size / dt_m,
                                                     "kernel_family": wm.last_timing()["kernel_family"]}
                wm.close()
            except Exception as e:  # noqa: BLE001
                out["single_evaluation_implicit_midpoint"] = {"error": str(e)[:200]}
            other = {}
            for ns in (4096, 6144):
                if ns == args.samples_per_gpu:
                    continue
                n2, w2, s2 = jq.cases.cnot3_ensemble(ns)
                jq.eval_f_g_grad(pcof, params, wa, n2, w2, True, shift=s2)
                t2 = wa.last_timing()
                other[str(ns)] = {"evals_per_s": ns / (t2["ms_total"] * 1e-3), "ms": t2["ms_total"], "kernel_family": t2["kernel_family"],
                                  "kernel_band": t2["kernel_band"]}
            out["other_batch_sizes"] = other
            mid = {}
            for ns in (512, 1024, 2048):      # mid-size ensembles: the latency staircase below one full round
                n2, w2, s2 = jq.cases.cnot3_ensemble(ns)
                jq.eval_f_g_grad(pcof, params, wa, n2, w2, True, shift=s2)
                jq.eval_f_g_grad(pcof, params, wa, n2, w2, True, shift=s2)
                t2 = wa.last_timing()
                mid[str(ns)] = {"evals_per_s": ns / (t2["ms_total"] * 1e-3), "ms": t2["ms_total"], "kernel_family": t2["kernel_family"]}
            out["mid_size_ensembles"] = mid
            _trace("latency / other batch sizes done")
            try:
                out["dense_operator"] = dense_operator_block(jq, L, pcof)
                # north_star's sentence (">= 40 % fp64-MFMA utilisation on the H x state-batch kernel") is about THIS formulation: its
                # roofline object is a sibling of `roofline` (which is the structured kernel the headline `value` runs on)
                out["roofline_dense"] = dict(out["dense_operator"]["roofline"], workload=out["dense_operator"]["workload"],
                                             evals_per_s=out["dense_operator"]["evals_per_s"])
            except Exception as e:  # noqa: BLE001  (a side measurement)
                out["dense_operator"] = {"error": repr(e)[:300]}
            _trace("dense operator done")
            # the other BASELINE.json configurations (parity-test cases, not bench lines): see baseline_configs_block
            try:
                out["baseline_configs"] = baseline_configs_block(jq, L, quick=False)
            except Exception as e:  # noqa: BLE001  (never let a side measurement take the bench line down)
                out["baseline_configs"] = {"error": repr(e)}
            _trace("baseline configs done")
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(params, pcof)
                _trace("cpu baseline done")
        print(json.dumps(out), flush=True)
    wa.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
