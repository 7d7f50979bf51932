"""ctypes wrapper of the CPU parity oracle (oracle/juqbox_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product
package (juqbox.jl_amd).  Parity status: pinned against every Stormer-Verlet golden of the
reference (tests/test_oracle_golden.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    so = os.path.join(_HERE, "libjuqbox_oracle.so")
    src = os.path.join(_HERE, "juqbox_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libjuqbox_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.jqo_create.restype = ctypes.c_void_p
        L.jqo_create.argtypes = [ctypes.c_int] * 5 + [ctypes.c_double] + [c_dp] * 8 + \
            [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
        L.jqo_destroy.argtypes = [ctypes.c_void_p]
        L.jqo_set_max_iter.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.jqo_set_uncoupled.argtypes = [ctypes.c_void_p, c_dp]
        L.jqo_set_target.argtypes = [ctypes.c_void_p, c_dp, c_dp]
        L.jqo_set_wdiag.argtypes = [ctypes.c_void_p, c_dp]
        L.jqo_set_wdense.argtypes = [ctypes.c_void_p, c_dp, c_dp]
        L.jqo_hconst.restype = c_dp
        L.jqo_hconst.argtypes = [ctypes.c_void_p]
        L.jqo_controls.argtypes = [ctypes.c_void_p, c_dp, ctypes.c_int, ctypes.c_double, c_dp]
        L.jqo_control_grad.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int, c_dp]
        L.jqo_traceobjgrad.argtypes = [ctypes.c_void_p, c_dp, ctypes.c_int, ctypes.c_int] + [c_dp] * 7
        L.jqo_traceobjgrad_imr.argtypes = [ctypes.c_void_p, c_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double] + [c_dp] * 6
        L.jqo_eval_f_g_grad.argtypes = [ctypes.c_void_p, c_dp, ctypes.c_int, c_dp, c_dp, ctypes.c_int, c_dp,
                                        ctypes.c_int, c_dp, c_dp, c_dp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def _f(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))


class Oracle:
    """One problem instance.  `params` is any object with the objparams field names."""

    def __init__(self, params, use_sparse=None):
        p = params
        self.Ntot, self.N, self.nsteps = p.Ntot, p.N, p.nsteps
        self.Ncoupled, self.Nfreq = p.Ncoupled, p.Nfreq
        nunc = getattr(p, "Nunc", 0)
        nctrl = p.Ncoupled
        if nunc:
            # uncoupled controls (src/evalobjgrad.jl:2373-2387; Ncoupled == 0, :176): Hunc_ops[q] goes to the symmetric slot
            # when it is symmetric, to the antisymmetric one otherwise (isSymm, :186-196); see jqo_set_uncoupled
            nctrl = nunc
            z = np.zeros(p.Ntot * p.Ntot)
            hs = np.concatenate([_f(h) if sym else z for h, sym in zip(p.Hunc_ops, p.isSymm)])
            ha = np.concatenate([z if sym else _f(h) for h, sym in zip(p.Hunc_ops, p.isSymm)])
        else:
            hs = np.concatenate([_f(h) for h in p.Hsym_ops]) if p.Ncoupled else np.zeros(0)
            ha = np.concatenate([_f(h) for h in p.Hanti_ops]) if p.Ncoupled else np.zeros(0)
        self.Ncoupled = nctrl
        sparse = p.use_sparse if use_sparse is None else use_sparse
        # leakage weights: a vector = the Diagonal default; a full matrix = use_custom_forbidden (src/evalobjgrad.jl:214-232), then
        # with wmat_imag next to it (Stormer-Verlet path only; parity-unpinned in the reference)
        dense_w = np.ndim(p.wmat_real) == 2
        wdiag = np.diag(np.asarray(p.wmat_real)).copy() if dense_w else p.wmat_real
        self._keep = [_f(p.Hconst), hs, ha, _f(p.Uinit), _f(p.Utarget_r), _f(p.Utarget_i),
                      _f(wdiag), _f(p.Cfreq[:nctrl, :])]
        self._wmat_real, self._wmat = _f(wdiag), _f(getattr(p, "wmat", wdiag))
        self.h = lib().jqo_create(p.Ntot, p.N, nctrl, p.Nfreq, p.nsteps, p.T,
                                  *[_p(a) for a in self._keep], p.objFuncType,
                                  p.linear_solver.solver_id, p.linear_solver.max_iter, p.linear_solver.tol,
                                  1 if sparse else 0)
        if nunc:
            self._rfreq = _f(p.Rfreq[:nunc])
            lib().jqo_set_uncoupled(self.h, _p(self._rfreq))
        if dense_w:
            wi = getattr(p, "wmat_imag", None)
            wi = _f(wi) if wi is not None and np.ndim(wi) == 2 else None
            lib().jqo_set_wdense(self.h, _p(_f(p.wmat_real)), _p(wi))

    def __del__(self):
        if getattr(self, "h", None):
            lib().jqo_destroy(self.h)
            self.h = None

    def set_max_iter(self, m):
        lib().jqo_set_max_iter(self.h, int(m))

    def controls(self, pcof, t):
        pcof = _f(pcof)
        out = np.zeros(2 * self.Ncoupled)
        rc = lib().jqo_controls(self.h, _p(pcof), pcof.size, float(t), _p(out))
        assert rc == 0
        return out

    def control_grad(self, ncoeff, t, func):
        g = np.zeros(ncoeff)
        rc = lib().jqo_control_grad(self.h, int(ncoeff), float(t), int(func), _p(g))
        assert rc == 0
        return g

    def traceobjgrad(self, pcof, evaladjoint=True, history=False, final_state=False):
        """Returns a dict: objfv, primaryobjf, secondaryobjf, traceInfidelity, totalgrad,
        infidelgrad, leakgrad [, history (complex [Ntot,N,nsteps+1]), final_state]."""
        pcof = _f(pcof)
        n = pcof.size
        out = np.zeros(4)
        tg, ig, lg = np.zeros(n), np.zeros(n), np.zeros(n)
        hr = hi = fs = None
        if history:
            hr = np.zeros(self.Ntot * self.N * (self.nsteps + 1))
            hi = np.zeros_like(hr)
        if final_state:
            fs = np.zeros(4 * self.Ntot * self.N)
        rc = lib().jqo_traceobjgrad(self.h, _p(pcof), n, 1 if evaladjoint else 0, _p(out), _p(tg), _p(ig), _p(lg),
                                    _p(hr), _p(hi), _p(fs))
        if rc == -1:
            raise ValueError("pcof must have an even number of elements >= 3*Nsig")
        if rc == -2:
            raise ValueError("DimensionMismatch: Inconsistent number of coefficients and size of parameter vector")
        res = dict(objfv=out[0], primaryobjf=out[1], secondaryobjf=out[2], traceInfidelity=out[3])
        if evaladjoint:
            res.update(totalgrad=tg, infidelgrad=ig, leakgrad=lg)
        if history:
            shp = (self.Ntot, self.N, self.nsteps + 1)
            res["history"] = hr.reshape(shp, order="F") + 1j * hi.reshape(shp, order="F")
        if final_state:
            res["final_state"] = fs.reshape((self.Ntot, self.N, 4), order="F")
        return res

    def traceobjgrad_imr(self, pcof, max_iter=100, tol=1e-12, evaladjoint=True, history=False):
        """Implicit-midpoint path (src/evalobjgrad.jl:1042-1481) with the JACOBI_SOLVER_M fixed-point solver."""
        pcof = _f(pcof)
        n = pcof.size
        out = np.zeros(4)
        tg, ig, lg = np.zeros(n), np.zeros(n), np.zeros(n)
        hr = hi = None
        if history:
            hr = np.zeros(self.Ntot * self.N * (self.nsteps + 1))
            hi = np.zeros_like(hr)
        lib().jqo_set_wdiag(self.h, _p(self._wmat))         # this path weights with params.wmat (src/evalobjgrad.jl:1147)
        try:
            rc = lib().jqo_traceobjgrad_imr(self.h, _p(pcof), n, 1 if evaladjoint else 0, int(max_iter), float(tol), _p(out),
                                            _p(tg), _p(ig), _p(lg), _p(hr), _p(hi))
        finally:
            lib().jqo_set_wdiag(self.h, _p(self._wmat_real))
        if rc != 0:
            raise ValueError("invalid pcof length (rc %d)" % rc)
        res = dict(objfv=out[0], primaryobjf=out[1], secondaryobjf=out[2], traceInfidelity=out[3])
        if evaladjoint:
            res.update(totalgrad=tg, infidelgrad=ig, leakgrad=lg)
        if history:
            shp = (self.Ntot, self.N, self.nsteps + 1)
            res["history"] = hr.reshape(shp, order="F") + 1j * hi.reshape(shp, order="F")
        return res

    def eval_f_g_grad(self, pcof, nodes, weights, shift, compute_adjoint=True):
        pcof = _f(pcof)
        nodes, weights, shift = _f(nodes), _f(weights), _f(shift)
        n = pcof.size
        out = np.zeros(2)
        ig, lg = np.zeros(n), np.zeros(n)
        rc = lib().jqo_eval_f_g_grad(self.h, _p(pcof), n, _p(nodes), _p(weights), nodes.size, _p(shift),
                                     1 if compute_adjoint else 0, _p(out), _p(ig), _p(lg))
        assert rc == 0, rc
        return dict(last_infidelity=out[0], last_leak=out[1], last_infidelity_grad=ig, last_leak_grad=lg)
