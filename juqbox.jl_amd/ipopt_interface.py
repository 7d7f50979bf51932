"""Mirror of the reference's optimiser glue around the hot path (src/ipopt_interface.jl:24-179):
eval_f_g_grad! (the risk-neutral ensemble = the multi-GPU sharding site) and the Ipopt callbacks
that consume its memoised results.  Ipopt itself stays outside (the callbacks have the reference's
signatures, so any L-BFGS driver can call them)."""
import numpy as np

from . import _lib
from .evalobjgrad import Working_Arrays_HIP, _f64, _ptr
from .setup_utils import tikhonov_grad, tikhonov_pen


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:  # torch absent: single process
        pass
    return None


def shard_bounds(nquad, rank, world):
    """Contiguous block partition of the nquad ensemble samples over `world` ranks."""
    base, rem = divmod(nquad, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_(vec, device=None):
    """ONE all-reduce (sum, fp64) of the packed result vector over all ranks (RCCL when the process
    group's backend is nccl; gloo in the CPU tests).  No-op without an initialised process group."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return vec
    import torch
    t = torch.from_numpy(vec)
    if dist.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    vec[:] = t.cpu().numpy()
    return vec


def _hip_shard_eval(pcof, params, wa, nodes, weights, shift, compute_adjoint):
    """Evaluate one shard of the ensemble on this rank's GPU: returns the packed partial sums
    [infidelity, leak, grad_infid(nCoeff), grad_leak(nCoeff)]."""
    if not isinstance(wa, Working_Arrays_HIP):
        raise TypeError("eval_f_g_grad: wa must be a Working_Arrays_HIP")
    L, h = _lib.load(), wa.handle
    n = pcof.size
    wa.sync_params()
    out2 = np.zeros(2)
    ig, lg = np.zeros(n), np.zeros(n)
    sh = _f64(shift) if shift is not None else None
    _lib.check(L.jq_eval_f_g_grad(h, _ptr(pcof), n, _ptr(nodes), _ptr(weights), nodes.size, _ptr(sh),
                                  1 if compute_adjoint else 0, _ptr(out2), _ptr(ig), _ptr(lg)), h)
    return np.concatenate([out2, ig, lg])


def eval_f_g_grad(pcof, params, wa, nodes=(0.0,), weights=(1.0,), compute_adjoint=True, shift=None,
                  distributed=True, _shard_eval=_hip_shard_eval):
    """eval_f_g_grad!(pcof, params, wa, nodes, weights, compute_adjoint) -- src/ipopt_interface.jl:24-70.

    All quadrature nodes are evaluated concurrently on the GPU as one batch of N*nquad columns; with an
    initialised torch.distributed process group the nodes are block-partitioned over the ranks and the
    packed result [infidelity, leak, grad_infid(nCoeff), grad_leak(nCoeff)] is summed with ONE
    all-reduce.  Results land in params.last_* exactly like the reference (:27-31, :48-59, :67-68).
    `_shard_eval` exists for the CPU (gloo) tests of the sharding logic; the product default is the
    HIP library and there is no CPU fallback."""
    pcof = _f64(pcof)
    n = pcof.size
    nodes = _f64(nodes)
    weights = _f64(weights)
    if nodes.size != weights.size:
        raise ValueError("nodes and weights must have the same length")
    dist = _dist() if distributed else None
    lo, hi = 0, nodes.size
    if dist is not None and dist.get_world_size() > 1:
        lo, hi = shard_bounds(nodes.size, dist.get_rank(), dist.get_world_size())
    packed = np.zeros(2 + 2 * n)
    if hi > lo:
        packed[:] = _shard_eval(pcof, params, wa, nodes[lo:hi].copy(), weights[lo:hi].copy(), shift, compute_adjoint)
    if dist is not None and dist.get_world_size() > 1:
        allreduce_sum_(packed)
    params.last_pcof = pcof.copy()
    params.last_infidelity = float(packed[0])
    params.last_leak = float(packed[1])
    if compute_adjoint:
        params.last_infidelity_grad = packed[2:2 + n].copy()
        params.last_leak_grad = packed[2 + n:].copy() if params.objFuncType != 1 else np.zeros(0)
    params.lastTraceInfidelity = params.last_infidelity
    params.lastLeakIntegral = params.last_leak
    return params.last_infidelity, params.last_leak


def _stale(pcof, params):
    # memoisation on ||pcof - last_pcof|| > 1e-15 (src/ipopt_interface.jl:83-84)
    last = params.last_pcof
    return last.size != np.size(pcof) or np.linalg.norm(np.asarray(pcof, dtype=np.float64) - last) > 1.0e-15


def eval_f_par(pcof, params, wa, nodes=(0.0,), weights=(1.0,)):
    """src/ipopt_interface.jl:77-99"""
    if _stale(pcof, params):
        eval_f_g_grad(pcof, params, wa, nodes, weights, True)
    f = params.last_infidelity + params.last_leak if params.objFuncType == 1 else params.last_infidelity
    prior = params.priorCoeffs if params.usingPriorCoeffs else None
    return f + tikhonov_pen(pcof, params.tik0, prior)


def eval_g_par(pcof, g, params, wa, nodes=(0.0,), weights=(1.0,)):
    """src/ipopt_interface.jl:104-118"""
    if _stale(pcof, params):
        eval_f_g_grad(pcof, params, wa, nodes, weights, True)
    g[0] = params.last_leak
    return g[0]


def eval_grad_f_par(pcof, grad_f, params, wa, nodes=(0.0,), weights=(1.0,)):
    """src/ipopt_interface.jl:124-148"""
    if _stale(pcof, params):
        eval_f_g_grad(pcof, params, wa, nodes, weights, True)
    grad_f[:] = params.last_infidelity_grad
    prior = params.priorCoeffs if params.usingPriorCoeffs else None
    wa.gr[:] = tikhonov_grad(pcof, params.tik0, prior)
    grad_f += wa.gr
    if params.save_pcof_hist:
        params.pcof_hist.append(np.array(pcof, dtype=np.float64))


def eval_jac_g_par(pcof, rows, cols, jac_g, params, wa, nodes=(0.0,), weights=(1.0,)):
    """src/ipopt_interface.jl:153-179 (including its quirk: when it has to recompute it returns
    without filling jac_g, :169-173)."""
    if jac_g is None:
        if len(rows) > 0:
            for i in range(len(pcof)):
                rows[i] = 1
                cols[i] = i + 1
        return
    if _stale(pcof, params):
        eval_f_g_grad(pcof, params, wa, nodes, weights, True)
        return
    jac_g[:] = params.last_leak_grad


def traceobj_sweep(pcof, params, wa, ep_vals, shift=None):
    """The loop of ep_plot (examples/Risk_Neutral/run_all.jl:6-32) as one batched call: returns an
    array [len(ep_vals), 4] = (objfv, primaryobjf, secondaryobjf, traceInfidelity) per perturbation."""
    L, h = _lib.load(), wa.handle
    pcof = _f64(pcof)
    ep = _f64(ep_vals)
    wa.sync_params()
    out = np.zeros(4 * ep.size)
    sh = _f64(shift) if shift is not None else None
    _lib.check(L.jq_traceobj_sweep(h, _ptr(pcof), pcof.size, _ptr(ep), ep.size, _ptr(sh), _ptr(out)), h)
    return out.reshape((ep.size, 4))
