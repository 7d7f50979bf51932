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


def _shard_bounds_py(nquad, rank, world):
    """jq_shard_bounds restated (juqbox_hip.hip): shard `rank` owns [lo, hi); the first nquad % world shards get one more."""
    nquad, rank, world = int(nquad), int(rank), int(world)
    if nquad < 0 or world < 1 or not 0 <= rank < world:
        raise ValueError("shard_bounds: need nquad >= 0, world >= 1, 0 <= rank < world")
    base, rem = divmod(nquad, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_bounds(nquad, rank, world):
    """Contiguous block partition of the nquad ensemble samples over `world` ranks: the library's own
    jq_shard_bounds (the partition a multi-device handle uses for its GPUs), so ranks and devices shard alike.
    Pure host arithmetic: where the library cannot be loaded (CPU-only sharding tests on a box without the build) the
    same rule is evaluated in Python (tests/test_host_logic.py pins the two against each other)."""
    import ctypes
    try:
        L = _lib.load()
    except (ImportError, OSError, AttributeError):
        return _shard_bounds_py(nquad, rank, world)
    lo, hi = ctypes.c_int32(), ctypes.c_int32()
    _lib.check(L.jq_shard_bounds(int(nquad), int(rank), int(world), ctypes.byref(lo), ctypes.byref(hi)))
    return lo.value, hi.value


def allreduce_sum_(vec):
    """ONE all-reduce (sum, fp64) of a packed host vector over all ranks (the gloo path of the CPU tests; the RCCL path
    keeps the vector on the device, _hip_shard_eval_dev).  No-op without an initialised process group."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return vec
    import torch
    t = torch.from_numpy(vec)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
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


def _hip_shard_eval_dev(pcof, params, wa, nodes, weights, shift, compute_adjoint):
    """The same evaluation for a job with one process per GPU over RCCL: the library leaves the packed partial sums
    on the device (jq_eval_f_g_grad_dev), torch.distributed all-reduces them in place, ONE copy brings the result back.
    A rank without a shard contributes zeros."""
    import torch
    dist = _dist()
    L, h = _lib.load(), wa.handle
    n = pcof.size
    wa.sync_params()
    t = torch.empty(2 + 2 * n, dtype=torch.float64, device=torch.device("cuda", wa.device))      # on the HANDLE's device
    sh = _f64(shift) if shift is not None else None
    import ctypes
    _lib.check(L.jq_eval_f_g_grad_dev(h, _ptr(pcof), n, _ptr(nodes), _ptr(weights), nodes.size, _ptr(sh),
                                      1 if compute_adjoint else 0, ctypes.c_void_p(t.data_ptr())), h)
    import time
    t0 = time.perf_counter()                      # (jq_eval_f_g_grad_dev returned after its stream was synchronised)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    res = t.cpu().numpy()                         # waits for the collective
    wa.last_allreduce_ms = (time.perf_counter() - t0) * 1e3
    return res


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
    # reset the memoised gradients like the reference (src/ipopt_interface.jl:27-31): a call with compute_adjoint = false
    # must not leave the gradients of an older pcof behind the new last_pcof
    params.last_infidelity_grad = np.zeros(n)
    params.last_leak_grad = np.zeros(n) if params.objFuncType != 1 else np.zeros(0)
    packed = np.zeros(2 + 2 * n)
    if dist is not None and dist.get_backend() == "nccl" and _shard_eval is _hip_shard_eval and wa.num_devices == 1:
        # one process per GPU (also with a single rank): partial sums stay on the device for the RCCL all-reduce
        packed[:] = _hip_shard_eval_dev(pcof, params, wa, nodes[lo:hi].copy(), weights[lo:hi].copy(), shift, compute_adjoint)
    else:
        if hi > lo:
            packed[:] = _shard_eval(pcof, params, wa, nodes[lo:hi].copy(), weights[lo:hi].copy(), shift, compute_adjoint)
        if dist is not None and dist.get_world_size() > 1:
            if dist.get_backend() == "nccl":      # (multi-device handle under an RCCL process group: collective on a device copy)
                import torch
                t = torch.from_numpy(packed).to(torch.device("cuda", wa.device))
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                packed[:] = t.cpu().numpy()
            else:
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


# ---------------------------------------------------------------------------------------------
# Optimiser loop on top of the callbacks.  The reference hands the callbacks to
# Ipopt (L-BFGS Hessian approximation, bound constraints, optionally the leakage as one inequality
# constraint); Ipopt is not available in this image, so run_optimizer drives the SAME callbacks with
# scipy's bound-constrained quasi-Newton methods.  Names, arguments, memoisation, convergence history and
# the early-stop thresholds follow src/ipopt_interface.jl:205-437; the iterates differ from Ipopt's
# (different line search / barrier), the objective and gradient they see do not.
def intermediate_par(alg_mod, iter_count, obj_value, inf_pr, inf_du, mu, d_norm, regularization_size,
                     alpha_du, alpha_pr, ls_trials, params):
    """src/ipopt_interface.jl:205-236: record the convergence history, stop on the thresholds."""
    if params.saveConvHist:
        params.objHist.append(obj_value)
        params.dualInfidelityHist.append(inf_du)
        params.primaryHist.append(params.lastTraceInfidelity)
        params.secondaryHist.append(params.lastLeakIntegral)
    if obj_value < params.objThreshold:
        if not params.quiet:
            print("Stopping because objective value = ", obj_value, " < threshold = ", params.objThreshold)
        return False
    if params.lastTraceInfidelity < params.traceInfidelityThreshold:
        if not params.quiet:
            print("Stopping because trace infidelity = ", params.lastTraceInfidelity, " < threshold = ",
                  params.traceInfidelityThreshold)
        return False
    return True


class OptimProblem:
    """What setup_ipopt_problem returns: callbacks + options (the fields an IpoptProblem carries)."""

    def __init__(self):
        self.x = None
        self.obj_val = None
        self.status = None
        self.n_iter = 0


def setup_ipopt_problem(params, wa, nCoeff, minCoeff, maxCoeff, maxIter=50, lbfgsMax=10, startFromScratch=True,
                        ipTol=1.0e-5, acceptTol=1.0e-5, acceptIter=15, nodes=(0.0,), weights=(1.0,),
                        jacob_approx="exact"):
    """src/ipopt_interface.jl:262-415."""
    minCoeff = np.asarray(minCoeff, dtype=np.float64)
    maxCoeff = np.asarray(maxCoeff, dtype=np.float64)
    if minCoeff.size != nCoeff or maxCoeff.size != nCoeff:
        raise ValueError("minCoeff and maxCoeff must have nCoeff elements")
    rng = np.random.default_rng()
    params.last_pcof = 1e9 * rng.random(nCoeff)              # :277-281: force the first evaluation
    params.last_infidelity_grad = 1e9 * rng.random(nCoeff)
    if params.objFuncType != 1:
        params.last_leak_grad = 1e9 * rng.random(nCoeff)
    nodes = np.asarray(nodes, dtype=np.float64)
    weights = np.asarray(weights, dtype=np.float64)

    prob = OptimProblem()
    prob.params, prob.wa, prob.nCoeff = params, wa, int(nCoeff)
    prob.x_L, prob.x_U = minCoeff, maxCoeff
    prob.eval_f = lambda pcof: eval_f_par(pcof, params, wa, nodes, weights)

    def _grad(pcof):
        g = np.zeros(nCoeff)
        eval_grad_f_par(pcof, g, params, wa, nodes, weights)
        return g
    prob.eval_grad_f = _grad
    if params.objFuncType == 3:                               # :299-306: leakage as an inequality constraint
        prob.m = 1
        prob.g_L, prob.g_U = np.array([-2e19]), np.array([params.leak_ubound])
    else:
        prob.m = 0
        prob.g_L, prob.g_U = np.zeros(0), np.zeros(0)

    def _g(pcof):
        g = np.zeros(1)
        eval_g_par(pcof, g, params, wa, nodes, weights)
        return g

    def _jac_g(pcof):
        jac = np.zeros(nCoeff)
        if _stale(pcof, params):     # the reference's callback returns unfilled in this case (:169-173); Ipopt
            eval_f_g_grad(pcof, params, wa, nodes, weights, True)   # always calls eval_g first -- here we recompute
        eval_jac_g_par(pcof, [], [], jac, params, wa, nodes, weights)
        return jac
    prob.eval_g, prob.eval_jac_g = _g, _jac_g
    prob.intermediate = lambda it, obj, inf_du: intermediate_par(0, it, obj, 0.0, inf_du, 0.0, 0.0, 0.0, 0.0, 0.0, 0,
                                                                 params)
    prob.options = dict(max_iter=int(maxIter), limited_memory_max_history=int(lbfgsMax), tol=float(ipTol),
                        acceptable_tol=float(acceptTol), acceptable_iter=int(acceptIter),
                        warm_start=not startFromScratch, jacobian_approximation=jacob_approx)
    if not params.quiet:
        print("Optimizer parameters: max # iterations = ", maxIter)
        print("Optimizer parameters: max history L-BFGS = ", lbfgsMax)
        print("Optimizer parameters: tol = ", ipTol)
    return prob


class _Stop(Exception):
    pass


def run_optimizer(prob, pcof0, baseName=""):
    """src/ipopt_interface.jl:417-437: optimise the control vector, optionally save it as <baseName>.jld2."""
    from scipy import optimize
    from .pcof_io import save_pcof
    params = prob.params
    x0 = np.clip(np.array(pcof0, dtype=np.float64), prob.x_L, prob.x_U)     # copy: pcof0 is not overwritten
    opt = prob.options
    bounds = optimize.Bounds(prob.x_L, prob.x_U)
    state = {"it": 0, "x": x0.copy()}

    def callback(xk, *_):
        state["it"] += 1
        state["x"] = np.array(xk, dtype=np.float64)
        obj = prob.eval_f(xk)                      # memoised: no extra propagation for an accepted iterate
        g = prob.eval_grad_f(xk)
        # projected-gradient norm as the dual infeasibility measure of the bound-constrained problem
        pg = np.where((xk <= prob.x_L) & (g > 0) | (xk >= prob.x_U) & (g < 0), 0.0, g)
        if not prob.intermediate(state["it"], obj, float(np.max(np.abs(pg))) if pg.size else 0.0):
            raise _Stop()

    if not params.quiet:
        print("*** Starting the optimization ***")
    try:
        if prob.m == 0:
            res = optimize.minimize(prob.eval_f, x0, jac=prob.eval_grad_f, method="L-BFGS-B", bounds=bounds,
                                    callback=callback,
                                    options=dict(maxiter=opt["max_iter"], maxcor=opt["limited_memory_max_history"],
                                                 ftol=0.0, gtol=opt["tol"]))
        else:
            cons = [dict(type="ineq", fun=lambda x: prob.g_U - prob.eval_g(x), jac=lambda x: -prob.eval_jac_g(x)[None, :])]
            res = optimize.minimize(prob.eval_f, x0, jac=prob.eval_grad_f, method="SLSQP", bounds=bounds,
                                    constraints=cons, callback=callback,
                                    options=dict(maxiter=opt["max_iter"], ftol=opt["tol"] * 1e-3))
        x, prob.status = np.array(res.x), str(res.message)
    except _Stop:
        x, prob.status = state["x"], "stopped by intermediate callback (threshold reached)"
    prob.x = x
    prob.obj_val = prob.eval_f(x)
    prob.n_iter = state["it"]
    if len(baseName) > 0:
        save_pcof(baseName + ".jld2", x)
        if not params.quiet:
            print("Saved B-spline parameters on binary jld2-file '%s.jld2'" % baseName)
    return x
