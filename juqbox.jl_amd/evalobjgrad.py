"""Host-side mirror of the reference's hot-path entry points over the C ABI:

  Working_Arrays_HIP(params, nCoeff)    the drop-in third working-array type:
                                        where the reference dispatches traceobjgrad on
                                        wa::Working_Arrays (src/evalobjgrad.jl:504) this type forwards
                                        to libjuqbox_hip.so.  Offers `wa.gr` like Working_Arrays (:400),
                                        which eval_grad_f_par uses (src/ipopt_interface.jl:139-141).
  traceobjgrad(pcof0, params, wa, verbose=False, evaladjoint=True)
                                        same argument order and return tuples as
                                        src/evalobjgrad.jl:504, :1027-1036.
"""
import contextlib
import ctypes
import threading

import numpy as np

from . import _lib
from .objparams import (JACOBI_SOLVER, JACOBI_SOLVER_M, NEUMANN_SOLVER, Implicit_Midpoint, Stormer_Verlet,
                        objparams)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))


def _ptr(a):
    return a.ctypes.data_as(_lib.c_dp) if a is not None else None


class _Csc:
    """A dense operator in the fields of Julia's SparseMatrixCSC{Float64,Int64} (1-based Int64 colptr / rowval) with the
    jq_csc descriptor pointing at them -- what the Julia binding passes for use_sparse = true problems."""

    def __init__(self, M):
        M = np.asarray(M, dtype=np.float64)
        n = M.shape[0]
        cols, rows, vals = [1], [], []
        for j in range(n):
            nz = np.nonzero(M[:, j])[0]
            rows.extend((nz + 1).tolist())
            vals.extend(M[nz, j].tolist())
            cols.append(len(rows) + 1)
        self.colptr = np.array(cols, dtype=np.int64)
        self.rowval = np.array(rows if rows else [1], dtype=np.int64)
        self.nzval = np.array(vals if vals else [0.0], dtype=np.float64)
        self.desc = _lib.jq_csc(n, n, self.colptr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                self.rowval.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _ptr(self.nzval))


def _csc_array(mats):
    keep = [_Csc(M) for M in mats]
    arr = (_lib.jq_csc * max(len(keep), 1))(*[k.desc for k in keep])
    return keep, arr


# ---- per-handle options (include/juqbox_hip.h: jq_create_opts / jq_set_option; the table of names is in INTEGRATION.md section 4) ----
_defaults = threading.local()


def _option_string(opts):
    return ",".join("%s=%d" % (k, int(v)) for k, v in opts.items())


def _norm_options(opts):
    """{"quad": 0} or the historic spelling {"JQ_QUAD": "0"} -> {"quad": 0}"""
    return {(k[3:].lower() if k.startswith("JQ_") else k): int(v) for k, v in (opts or {}).items()}


@contextlib.contextmanager
def options(**opts):
    """Default options for the handles CREATED inside the `with` block (tests, bisection):
        with jq.options(quad=0, coop_max=0):
            wa = jq.Working_Arrays_HIP(params, nCoeff)
    They are options of those handles from then on (nothing is read from the environment); an explicit `options=` argument of the
    constructor is applied on top.  Nested blocks add up.  Options of an existing handle: wa.set_option(name, value)."""
    prev = dict(getattr(_defaults, "opts", {}))
    merged = dict(prev)
    merged.update(_norm_options(opts))
    _defaults.opts = merged
    try:
        yield
    finally:
        _defaults.opts = prev


class Working_Arrays_HIP:
    """Owns the device handle for one `objparams`.  Mutable fields of `params` that scripts change
    after construction (Hconst, wmat_real, Utarget_r/i, linear_solver.max_iter) are re-synchronised
    with the device at every call, so `params` stays the single source of truth like in the reference."""

    INTEGRATOR = Stormer_Verlet
    SOLVERS = (NEUMANN_SOLVER, JACOBI_SOLVER)

    def _weights(self, p):
        """leakage weights of this path: params.wmat_real (src/evalobjgrad.jl:583) -- a vector (the Diagonal default) or, with
        use_custom_forbidden (:214-232), a full matrix next to params.wmat_imag: then (wmat_real, wmat_imag) stacked"""
        wi = getattr(p, "wmat_imag", None)
        if np.ndim(p.wmat_real) == 2:
            wi = np.zeros_like(p.wmat_real) if wi is None or np.ndim(wi) != 2 else wi      # (Diagonal(zeros(Ntot)), :236)
            return np.concatenate([_f64(p.wmat_real), _f64(wi)])
        if wi is not None and np.any(np.asarray(wi) != 0):
            # a Diagonal wmat_real next to a non-zero wmat_imag: the Julia binding passes both as full matrices (a full wmat_imag) or
            # refuses (a Diagonal one is not a Hermitian weight); dropping wmat_imag silently would evaluate something else
            if np.ndim(wi) == 2:
                return np.concatenate([_f64(np.diag(np.asarray(p.wmat_real, dtype=np.float64))), _f64(wi)])
            raise ValueError("a non-zero Diagonal wmat_imag is not a Hermitian leakage weight (julia/hip_backend.jl refuses it too)")
        return _f64(p.wmat_real)

    def _push_weights(self, w):
        L, p, h = _lib.load(), self.params, self.handle
        if w.size == p.Ntot:
            _lib.check(L.jq_update_wmat_diag(h, _ptr(w)), h)
        elif w.size == 2 * p.Ntot * p.Ntot:
            wr, wi = w[:p.Ntot * p.Ntot].copy(), w[p.Ntot * p.Ntot:].copy()
            _lib.check(L.jq_update_wmat(h, _ptr(wr), _ptr(wi)), h)
        else:
            raise ValueError("wmat_real must be a vector of length Ntot (Diagonal) or an Ntot x Ntot matrix")

    def __init__(self, params: objparams, nCoeff: int, devices=None, csc=None, options=None):
        """devices: None = the current HIP device (one process per GPU); an int n or a list of device ids = ONE process
        driving several GPUs (jq_create_multi: the ensemble of eval_f_g_grad is sharded over them and summed with one
        RCCL all-reduce inside the library).
        csc: hand the operators over in sparse (SparseMatrixCSC) storage like the Julia binding does for use_sparse = true
        problems (default: params.use_sparse); the results are bit-identical to the dense form.
        options: {name: value} for jq_create_opts (on top of the defaults of an enclosing `with options(...)` block)."""
        L = _lib.load()
        if params.linear_solver.solver_id not in self.SOLVERS:
            raise ValueError("Please specify a supported linear solver")
        self.params = params
        self.nCoeff = int(nCoeff)
        self.gr = np.zeros(self.nCoeff)            # Working_Arrays.gr (src/evalobjgrad.jl:400)
        p = params
        hs = np.concatenate([_f64(h) for h in p.Hsym_ops]) if p.Ncoupled else np.zeros(1)
        ha = np.concatenate([_f64(h) for h in p.Hanti_ops]) if p.Ncoupled else np.zeros(1)
        self._hconst = _f64(p.Hconst).copy()
        w0 = self._weights(p).copy()
        self._wd = w0 if w0.size == p.Ntot else np.zeros(p.Ntot)      # (full weights follow the creation: jq_update_wmat)
        self._utr = _f64(p.Utarget_r).copy()
        self._uti = _f64(p.Utarget_i).copy()
        self._m = int(p.linear_solver.max_iter) if self.INTEGRATOR == Stormer_Verlet else 0
        self._solver = None
        nunc = getattr(p, "Nunc", 0)
        hu = np.concatenate([_f64(h) for h in p.Hunc_ops]) if nunc else None
        rf = _f64(p.Rfreq[:nunc]) if nunc else None
        keep = [self._hconst, hs, ha, _f64(p.Uinit), self._utr, self._uti, self._wd, _f64(p.Cfreq[:p.Ncoupled + nunc, :]), hu, rf]
        ptrs = [_ptr(a) for a in keep]
        sparse_args = [None, None, None]
        use_csc = bool(getattr(p, "use_sparse", False)) if csc is None else bool(csc)
        if use_csc and not nunc:
            k0, a0 = _csc_array([p.Hconst])
            ks, as_ = _csc_array(p.Hsym_ops)
            ka, aa = _csc_array(p.Hanti_ops)
            keep += [k0, a0, ks, as_, ka, aa]
            ptrs[0] = ptrs[1] = ptrs[2] = None
            sparse_args = [a0, as_, aa]
        self._csc = use_csc and not nunc
        prob = _lib.jq_problem(p.Ntot, p.N, p.Ncoupled, p.Nfreq, p.nsteps, self._m, p.objFuncType, nunc, p.T, *ptrs, *sparse_args)
        h = ctypes.c_void_p()
        opts = dict(getattr(_defaults, "opts", {}))
        opts.update(_norm_options(options))
        self.options = opts
        ostr = _option_string(opts).encode() if opts else None
        if devices is None:
            opts.pop("multi_same_device", None)
            ostr = _option_string(opts).encode() if opts else None
            rc = L.jq_create_opts(ctypes.byref(prob), ostr, ctypes.byref(h))
        else:
            devs = list(range(devices)) if isinstance(devices, int) else [int(d) for d in devices]
            arr = (ctypes.c_int32 * len(devs))(*devs)
            rc = L.jq_create_multi_opts(ctypes.byref(prob), arr, len(devs), ostr, ctypes.byref(h))
        if rc != _lib.JQ_OK:
            msg = L.jq_last_error(None)
            raise _lib.JuqboxHipError(rc, msg.decode() if msg else "?")
        self.handle = h
        self.num_devices = L.jq_num_devices(h)
        self.device = L.jq_handle_device(h)      # HIP device the handle is bound to (first one of a multi-device handle)
        self.last_allreduce_ms = 0.0             # one process per GPU: wall time of the caller's all-reduce (ipopt_interface.py)
        if w0.size != p.Ntot:
            try:
                self._push_weights(w0)
            except Exception:
                self.close()
                raise
            self._wd = w0

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().jq_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync_params(self):
        """Push post-construction mutations of `params` to the device (only what changed)."""
        L, p, h = _lib.load(), self.params, self.handle
        ls = p.linear_solver
        if ls.solver_id not in self.SOLVERS:
            raise ValueError("Please specify a supported linear solver")
        key = (int(ls.solver_id), int(ls.max_iter), float(ls.tol))
        # Diagonal weights go in BEFORE the solver / integrator, full weights AFTER it: full weights exist with the Neumann solver only,
        # so (full weights, Neumann) <-> (Diagonal, Jacobi) is a valid switch in one step in either direction (as julia/hip_backend.jl)
        wd = self._weights(p)
        if wd.size == p.Ntot and (wd.size != self._wd.size or not np.array_equal(wd, self._wd)):
            self._push_weights(wd)
            self._wd = wd.copy()
        if key != self._solver:
            if self.INTEGRATOR == Stormer_Verlet:
                _lib.check(L.jq_set_linear_solver(h, key[0], key[1], key[2]), h)
                self._m = key[1]
            else:
                _lib.check(L.jq_set_integrator(h, Implicit_Midpoint, key[1], key[2]), h)
            self._solver = key
        hc = _f64(p.Hconst)
        if not np.array_equal(hc, self._hconst):
            if self._csc:
                k0 = _Csc(p.Hconst)
                _lib.check(L.jq_update_hconst_csc(h, ctypes.byref(k0.desc)), h)
            else:
                _lib.check(L.jq_update_hconst(h, _ptr(hc)), h)
            self._hconst = hc.copy()
        if wd.size != self._wd.size or not np.array_equal(wd, self._wd):      # (full weights)
            self._push_weights(wd)
            self._wd = wd.copy()
        utr, uti = _f64(p.Utarget_r), _f64(p.Utarget_i)
        if not (np.array_equal(utr, self._utr) and np.array_equal(uti, self._uti)):
            _lib.check(L.jq_update_target(h, _ptr(utr), _ptr(uti)), h)
            self._utr, self._uti = utr.copy(), uti.copy()

    def set_option(self, name, value=None):
        """jq_set_option: change one option of this handle (value None: back to "not set"); plan-shaping options re-plan it"""
        name = name[3:].lower() if name.startswith("JQ_") else name
        v = _lib.JQ_OPTION_DEFAULT if value is None else int(value)
        _lib.check(_lib.load().jq_set_option(self.handle, name.encode(), v), self.handle)

    def get_option(self, name):
        v = ctypes.c_int64()
        rc = _lib.load().jq_get_option(self.handle, name.encode(), ctypes.byref(v))
        if rc != _lib.JQ_OK:
            raise KeyError(name)
        return None if v.value == _lib.JQ_OPTION_DEFAULT else v.value

    @property
    def num_compute_units(self):
        return _lib.load().jq_num_compute_units(self.handle)

    @property
    def rccl_world_size(self):
        return _lib.load().jq_rccl_world_size(self.handle)

    def plan_info(self):
        """jq_plan_info: structure found in the operators, control groups, batch-size thresholds of the kernel families (dict)"""
        import json
        L = _lib.load()
        n = L.jq_plan_info(self.handle, None, 0)
        if n < 0:
            _lib.check(n, self.handle)
        buf = ctypes.create_string_buffer(n + 1)
        L.jq_plan_info(self.handle, buf, n + 1)
        return json.loads(buf.value.decode())

    def last_timing(self):
        t = _lib.jq_timing()
        _lib.check(_lib.load().jq_last_timing(self.handle, ctypes.byref(t)), self.handle)
        return {k: getattr(t, k) for k, _ in _lib.jq_timing._fields_}


class Working_Arrays_M_HIP(Working_Arrays_HIP):
    """Working_Arrays_M (src/evalobjgrad.jl:445-500): selects the IMPLICIT-MIDPOINT method of traceobjgrad
    (:1042-1481) the way the reference does, by the type of `wa`.  `params.linear_solver` must be
    lsolver_object(solver=JACOBI_SOLVER_M, max_iter=..., tol=...) (test/runtests.jl:70); the leakage weights of this
    path are params.wmat (:1147), not params.wmat_real.  Device support: Ntot <= 96 (dense 96 x 96 operators excepted),
    N <= 16."""
    INTEGRATOR = Implicit_Midpoint
    SOLVERS = (JACOBI_SOLVER_M,)

    def _weights(self, p):
        return _f64(p.wmat)


def traceobjgrad(pcof0, params: objparams, wa: Working_Arrays_HIP, verbose: bool = False, evaladjoint: bool = True):
    """traceobjgrad(pcof0, params, wa, verbose, evaladjoint) -- src/evalobjgrad.jl:504-1038.

    evaladjoint (not verbose): (objfv, totalgrad, primaryobjf, secondaryobjf, traceInfidelity,
                                infidelgrad, leakgrad)                                     (:1033)
    neither:                   (objfv, primaryobjf, secondaryobjf)                         (:1035)
    verbose (not evaladjoint): (objfv, unitaryhistory[Ntot,N,nsteps+1] complex, fidelity)  (:1031)
    verbose and evaladjoint (the reference's forward-sensitivity self check, :1028) is out of scope.
    objfv excludes the Tikhonov term (added by the Ipopt callbacks, src/ipopt_interface.jl:96-98).
    """
    if not isinstance(wa, Working_Arrays_HIP):
        raise TypeError("traceobjgrad: wa must be a Working_Arrays_HIP")
    if wa.params is not params:
        raise ValueError("traceobjgrad: wa was allocated for a different objparams")
    if verbose and evaladjoint:
        raise NotImplementedError("verbose && evaladjoint (forward gradient self-check) is not accelerated")
    L, h = _lib.load(), wa.handle
    pcof = _f64(pcof0)
    n = pcof.size
    wa.sync_params()
    out4 = np.zeros(4)
    if verbose:
        shp = (params.Ntot, params.N, params.nsteps + 1)
        ur = np.zeros(int(np.prod(shp)))
        ui = np.zeros_like(ur)
        # history and objective from ONE forward sweep, like the reference's single pass
        _lib.check(L.jq_traceobj_verbose(h, _ptr(pcof), n, _ptr(out4), _ptr(ur), _ptr(ui)), h)
        hist = ur.reshape(shp, order="F") + 1j * ui.reshape(shp, order="F")
        return out4[0], hist, 1.0 - out4[3]
    if evaladjoint:
        tg, ig, lg = np.zeros(n), np.zeros(n), np.zeros(n)
        _lib.check(L.jq_traceobjgrad(h, _ptr(pcof), n, 1, _ptr(out4), _ptr(tg), _ptr(ig), _ptr(lg)), h)
        if params.objFuncType == 1:
            lg = np.zeros(0)      # the reference returns an empty leakgrad here (:808, :948-952)
        return out4[0], tg, out4[1], out4[2], out4[3], ig, lg
    _lib.check(L.jq_traceobjgrad(h, _ptr(pcof), n, 0, _ptr(out4), None, None, None), h)
    return out4[0], out4[1], out4[2]
