"""CPU, world_size 2, gloo: the multi-rank path of eval_f_g_grad (block-partitioned ensemble + ONE
all-reduce of the packed sums).  The per-shard evaluator is the CPU oracle here (the checker standing
in for the GPU call); the sharding, packing and collective are the product code."""
import os
import sys

import numpy as np
import pytest
from conftest import ROOT, case_inputs

torch = pytest.importorskip("torch")


def _oracle_shard(pcof, params, wa, nodes, weights, shift, adj):
    from oracle.oracle import Oracle
    sh = shift if shift is not None else params.shift_weights_reference()
    r = Oracle(params).eval_f_g_grad(pcof, nodes, weights, sh, adj)
    return np.concatenate([[r["last_infidelity"], r["last_leak"]], r["last_infidelity_grad"], r["last_leak_grad"]])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import juqbox_jl_amd.ipopt_interface as ii
    from conftest import case_inputs as ci
    params, info, pcof, _ = ci("swap02")
    x, w = np.polynomial.legendre.leggauss(5)
    nodes, weights = x * 0.5 * 0.1, w * 0.5
    ii.eval_f_g_grad(pcof, params, None, nodes, weights, True, _shard_eval=_oracle_shard)
    q.put((rank, params.last_infidelity, params.last_leak, params.last_infidelity_grad.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ensemble_matches_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    params, info, pcof, _ = case_inputs("swap02")
    x, w = np.polynomial.legendre.leggauss(5)
    nodes, weights = x * 0.5 * 0.1, w * 0.5
    full = _oracle_shard(pcof, params, None, nodes, weights, None, True)
    n = pcof.size
    for rank, inf, leak, grad in res:
        assert abs(inf - full[0]) < 1e-13
        assert abs(leak - full[1]) < 1e-15
        assert np.linalg.norm(grad - full[2:2 + n]) < 1e-12 * np.linalg.norm(full[2:2 + n])
    # both ranks hold identical (all-reduced) results
    assert res[0][1] == res[1][1] and np.array_equal(res[0][3], res[1][3])
