#!/usr/bin/env python3
"""bench.py -- traceobjgrad evals/sec (forward + discrete adjoint) at the cnot3 Hilbert dimension.

One "step" = one eval_f_g_grad pass of the hot path over the rank's batch of ensemble samples
(every sample is one full traceobjgrad evaluation of test/cases/cnot3-setup.jl: Ntot=96, N=4,
32 386 Stormer-Verlet steps, 6 Neumann terms, golden pcof).  Inputs (operators, pcof, ensemble
nodes) are resident in HBM when the timed region starts.  Weak scaling: every rank (GPU) gets
--samples-per-gpu samples; the four weighted sums are combined with ONE all-reduce (RCCL).

Prints ONE JSON line on rank 0 (see the contract in the repository brief)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak: 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz
                               # (v_mfma_f64_16x16x4_f64 issues every 64 clk; measured 75.4 TF, probes/)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    # 3072 samples = 768 slabs of 16 columns = three slabs on each of the 256 CUs: one round of the kernels that are fastest
    # for large ensembles (quad layout, 12 waves per workgroup).  Other sizes run too (4096 = one round of the slab kernels,
    # reported below as `other_batch_sizes`); the library picks the kernels per batch size.
    ap.add_argument("--samples-per-gpu", type=int, default=int(os.environ.get("JQ_BENCH_SAMPLES", "3072")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import juqbox_jl_amd as jq
    from juqbox_jl_amd import _lib
    from juqbox_jl_amd.ipopt_interface import shard_bounds

    L = _lib.load()
    torch.cuda.set_device(local_rank)
    _lib.check(L.jq_set_device(local_rank))
    dist = None
    if world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    params, info = jq.cases.cnot3()
    pcof = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "cnot3.json")))["pcof0"])
    nsamples_total = args.samples_per_gpu * world
    nodes, weights, shift = jq.cases.cnot3_ensemble(nsamples_total)
    wa = jq.Working_Arrays_HIP(params, pcof.size)

    def step():
        jq.eval_f_g_grad(pcof, params, wa, nodes, weights, True, shift=shift)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    prop_ms = bwd_ms = fwd_ms = 0.0
    nb = nf = 0
    mfma = 0
    tm = {}
    for _ in range(args.steps):
        step()
        tm = wa.last_timing()
        prop_ms += tm["ms_propagate"]
        bwd_ms += tm["ms_backward"]
        fwd_ms += tm["ms_forward"]
        nb += tm["n_backward_launches"]
        nf += tm["n_forward_launches"]
        mfma += tm["mfma_executed"]
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        evals = nsamples_total * args.steps
        value = evals / elapsed
        Ntot, N, Nc, m, nsteps = params.Ntot, params.N, params.Ncoupled, params.linear_solver.max_iter, params.nsteps
        lo, hi = shard_bounds(nsamples_total, 0, world)
        svts_rank = (hi - lo) * N * nsteps * args.steps
        f_bwd = 2.0 * Ntot * Ntot * (2 * (9 + 2 * m) + 7 * Nc)                 # algorithmic dense FLOP / SVTS, backward sweep
        f_fwd = 2.0 * Ntot * Ntot * (9 + 2 * m)
        # dominant kernel = k_backward<6>: algorithmic FLOPs per launch / average launch duration (HIP events
        # recorded on the library's stream around every launch, jq_last_timing)
        flops_per_launch = f_bwd * svts_rank / max(nb, 1)
        avg_launch_s = bwd_ms * 1e-3 / max(nb, 1)
        achieved = flops_per_launch / avg_launch_s / 1e12
        # HBM bytes per k_backward launch from the PMC passes kept under profiles/ (FETCH_SIZE x2 gfx950 correction
        # + WRITE_SIZE, separate rocprofv3 --pmc runs of this same command; null if not measured for this build)
        # the library reports which propagator family / instantiation ran (jq_timing.kernel_*)
        fam = {0: "k_backward", 1: "k_backward_coop", 2: "k_backward_lane", 3: "k_backward_rowlane", 6: "k_backward"}.get(tm.get("kernel_family", 0))
        kname = "%s<%d, %d>" % (fam, tm.get("kernel_size", 0), tm.get("kernel_band", 0))
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
            traffic = tj["kernels"][kname]["hbm_bytes_per_launch"]
        except Exception:
            pass
        band = tm.get("kernel_band")
        band_note = {9: " (band 9 = block tridiagonal with diagonal off-diagonal blocks)",
                     8: " (band 8 = 4x4 diagonal blocks on v_mfma_f64_4x4x4 + diagonal couplings on DPP FMAs)",
                     7: " (band 7 = the band-8 product in the quad layout: four columns per wave, a 16-row block per register; "
                        "%d waves per workgroup)" % (4 * max(1, round(args.samples_per_gpu * N / 16 / 256)) if args.samples_per_gpu * N / 16 <= 768 else 12)}.get(band, "")
        # arithmetic the kernels really execute (they skip the structural zeros the dense count includes): matrix pipe
        # from the library's MFMA count; for band 8 also the coupling FMAs of the products (6 NT + 8 (NT - 1) v_fma_f64
        # of 64 lanes per product, one product per 4 NT of the 512-FLOP MFMAs)
        NT = (Ntot + 15) // 16
        executed_mfma = mfma * 2048.0
        fma_per_product = (6 * NT + 8 * (NT - 1)) if band == 8 else 4 * (4 * NT - 2) if band == 7 else 0     # per slab of 16 columns
        executed_fma = (mfma * 4.0 / (4 * NT)) * fma_per_product * 128.0
        roofline = {"bound": "mfma", "kernel": kname + band_note,
                    "achieved_definition": "dense-contraction FLOPs of SURVEY.md 8(d) (2 Ntot^2 per product and column: what a dense "
                                           "formulation computes) / HIP-event time of the kernel; the kernels skip the structural zeros "
                                           "of the operators, so this exceeds the matrix peak (frac > 1) -- executed_* is the arithmetic "
                                           "really issued",
                    "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch (PMC)",
                    "mfma_pipe_util": mfma * 2048.0 / (prop_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                    "launches": int(nb), "avg_launch_ms": avg_launch_s * 1e3,
                    "all_propagators_tflops": (f_bwd + f_fwd) * svts_rank / (prop_ms * 1e-3) / 1e12,
                    "executed_mfma_tflops": executed_mfma / (prop_ms * 1e-3) / 1e12,
                    "executed_valu_fma_tflops": executed_fma / (prop_ms * 1e-3) / 1e12,
                    "executed_frac_of_fp64_peak": (executed_mfma + executed_fma) / (prop_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                    "forward_ms": fwd_ms, "backward_ms": bwd_ms}
        out = {"metric": "traceobjgrad evals/sec (fwd+adjoint), cnot3 Hilbert dim", "value": value,
               "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "cnot3 (test/cases/cnot3-setup.jl: Ntot=96, N=4, nsteps=32386, 6 Neumann terms, "
                                      "golden pcof) x risk-neutral ensemble of %d samples per GPU" % args.samples_per_gpu,
                          "samples_per_gpu": args.samples_per_gpu, "columns_per_gpu": args.samples_per_gpu * N,
                          "svts_per_step_all_gpus": nsamples_total * N * nsteps, "parallelism": "ensemble-dp%d" % world},
               "svts_per_s": nsamples_total * N * nsteps * args.steps / elapsed,
               "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            # outside the timed region: the latency of ONE evaluation (what an Ipopt iteration of the reference waits for;
            # quad-layout kernels) next to the CPU figure below -- not part of `value`
            jq.traceobjgrad(pcof, params, wa, False, True)
            t1 = time.perf_counter()
            jq.traceobjgrad(pcof, params, wa, False, True)
            torch.cuda.synchronize()
            ts = wa.last_timing()
            out["single_evaluation"] = {"seconds": time.perf_counter() - t1, "ms_propagate": ts["ms_propagate"],
                                        "kernel_family": ts["kernel_family"], "kernel_band": ts["kernel_band"]}
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
            from oracle.oracle import Oracle
            orc = Oracle(params)                      # sparse products like the reference's use_sparse=true
            nrep = 2
            t1 = time.perf_counter()
            for _ in range(nrep):
                orc.traceobjgrad(pcof)
            tc = (time.perf_counter() - t1) / nrep
            out["cpu_baseline"] = {"value": 1.0 / tc, "unit": "evals/s", "cores": 1, "kind": "port",
                                   "sample": "%d x one cnot3 traceobjgrad (1 sample = 4 columns x 32386 steps), C restatement "
                                             "of the reference's sparse Stormer-Verlet path, single thread like the reference" % nrep,
                                   "seconds_per_eval": tc, "host_cores_available": os.cpu_count()}
        print(json.dumps(out), flush=True)
    wa.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
