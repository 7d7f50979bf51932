"""Problem definition: Python mirror of the reference's `objparams` / `lsolver_object` data contract
(src/evalobjgrad.jl:53-345, src/linear_solvers.jl:28-78).

Only the fields the Stormer-Verlet hot path consumes are kept; the
field names are the reference's.  Like the reference struct it is a plain mutable object: set-up
scripts overwrite `wmat_real`, `tik0`, `linear_solver.max_iter`, ... after construction, and
`Working_Arrays_HIP` re-reads them at every call.
"""
import numpy as np

from . import setup_utils

Stormer_Verlet = 1      # src/evalobjgrad.jl:1
Implicit_Midpoint = 2   # src/evalobjgrad.jl:2 (accelerated for Ntot <= 16: Working_Arrays_M_HIP)

NEUMANN_SOLVER = 1      # src/linear_solvers.jl:5
JACOBI_SOLVER = 2       # src/linear_solvers.jl:6
JACOBI_SOLVER_M = 4     # src/linear_solvers.jl:8 (fixed-point solver of the implicit-midpoint step)


class lsolver_object:
    """src/linear_solvers.jl:28-65: solver id, max_iter (Neumann terms), tol (Jacobi only)."""

    def __init__(self, tol=1e-10, max_iter=3, nrhs=1, solver=NEUMANN_SOLVER):
        if solver == JACOBI_SOLVER:
            tol = tol * np.sqrt(nrhs)           # :40
            self.solver_name = "Jacobi"
        elif solver == NEUMANN_SOLVER:
            self.solver_name = "Neumann"
        elif solver == JACOBI_SOLVER_M:
            self.solver_name = "Jacobi from Implicit Midpoint"            # :52-55 (tol is NOT scaled by sqrt(nrhs))
        else:
            # GAUSSIAN_ELIM_SOLVER belongs to a path that is out of scope here
            raise ValueError("Please specify a supported linear solver")   # :59
        self.tol = float(tol)
        self.max_iter = int(max_iter)
        self.solver_id = int(solver)


class objparams:
    """objparams(Ne, Ng, T, nsteps; Uinit, Utarget, Cfreq, Rfreq, Hconst, Hsym_ops, Hanti_ops, ...)
    (constructor: src/evalobjgrad.jl:152-343).  Matrices are stored dense, column-major float64;
    `use_sparse` is remembered only as a hint (the device path keeps dense MFMA tiles and skips
    all-zero tiles; the CPU oracle uses it to pick the reference's sparse product)."""

    def __init__(self, Ne, Ng, T, nsteps, *, Uinit, Utarget, Cfreq, Rfreq, Hconst,
                 Hsym_ops=(), Hanti_ops=(), Hunc_ops=(), objFuncType=1, leak_ubound=1.0e-3,
                 wmatScale=1.0, use_sparse=False, linear_solver=None, Integrator=Stormer_Verlet,
                 use_custom_forbidden=False, forb_states=None, forb_weights=None):
        self.Ne = [int(x) for x in Ne]
        self.Ng = [int(x) for x in Ng]
        self.Nt = [a + b for a, b in zip(self.Ne, self.Ng)]
        self.Nosc = len(self.Ne)
        self.N = int(np.prod(self.Ne))
        self.Ntot = int(np.prod(self.Nt))
        self.Nguard = self.Ntot - self.N
        self.T = float(T)
        self.nsteps = int(nsteps)

        Cfreq = np.asfortranarray(np.asarray(Cfreq, dtype=np.float64))
        if Cfreq.ndim != 2:
            raise ValueError("Cfreq must be a matrix of size [Nctrl, Nfreq]")
        self.Nfreq = Cfreq.shape[1]
        self.Cfreq = Cfreq
        self.Rfreq = np.asarray(Rfreq, dtype=np.float64)

        self.Ncoupled = len(Hsym_ops)
        Nanti = len(Hanti_ops)
        self.Nunc = len(Hunc_ops)
        assert self.Ncoupled == 0 or self.Nunc == 0                               # :176
        assert len(self.Rfreq) >= self.Ncoupled + self.Nunc                      # :177
        # uncoupled controls (lab-frame evaluation of a pulse: KS! :2373-2387): every operator must be symmetric (-> K) or
        # antisymmetric (-> S), :186-196.  Parity-unpinned branch of the reference.
        self.Hunc_ops = [np.asfortranarray(np.asarray(h, dtype=np.float64)).copy(order="F") for h in Hunc_ops]
        self.isSymm = []
        for h in self.Hunc_ops:
            if np.array_equal(h, h.T):
                self.isSymm.append(True)
            elif np.linalg.norm(h + h.T) < 1e-15:
                self.isSymm.append(False)
            else:
                raise ValueError("Uncoupled Hamiltonian is not symmetric or anti-symmetric. This functionality is not "
                                 "currently supported.")                           # ArgumentError, :195
        tz = (self.Ntot, self.N)
        Uinit = np.asarray(Uinit, dtype=np.float64)
        Utarget = np.asarray(Utarget, dtype=np.complex128)
        assert Uinit.shape == tz, "size(Uinit) must be (Ntot, N)"                 # :181
        assert Utarget.shape == tz, "size(Utarget) must be (Ntot, N)"             # :182
        assert self.Ncoupled == Nanti, "Ncoupled == Nanti"                        # :243
        if Cfreq.shape[0] < self.Ncoupled + self.Nunc:
            raise ValueError("Cfreq needs one row per control Hamiltonian")

        self.Uinit = np.asfortranarray(Uinit)
        self.Utarget_r = np.asfortranarray(Utarget.real.copy())
        self.Utarget_i = np.asfortranarray(Utarget.imag.copy())
        self.use_bcarrier = True      # :208
        self.kpar = 1                 # :205
        self.tik0 = 0.01              # :202
        self.pFidType = 2             # :164

        nt = self.Ntot
        self.Hconst = np.asfortranarray(np.asarray(Hconst, dtype=np.float64)).copy(order="F")
        assert self.Hconst.shape == (nt, nt)
        self.Hsym_ops = [np.asfortranarray(np.asarray(h, dtype=np.float64)).copy(order="F") for h in Hsym_ops]
        self.Hanti_ops = [np.asfortranarray(np.asarray(h, dtype=np.float64)).copy(order="F") for h in Hanti_ops]
        for h in self.Hsym_ops + self.Hanti_ops:
            assert h.shape == (nt, nt)
        self.use_sparse = bool(use_sparse)

        # leakage weights.  Default: Diagonal (stored as the vector of its diagonal).  use_custom_forbidden (:214-232; parity-unpinned
        # in the reference: no test or example uses it): W = sum_k forb_weights[k] f_k f_k^H as FULL matrices wmat_real / wmat_imag
        # (W[i,j] += w conj(f_j) f_i), read by the Stormer-Verlet path only -- the implicit-midpoint path weights with `wmat`.
        self.wmat = wmatScale * setup_utils.wmatsetup(self.Ne, self.Ng)          # :211
        if use_custom_forbidden:
            fs = np.asarray(forb_states, dtype=np.complex128)
            fw = np.asarray(forb_weights, dtype=np.float64).ravel()
            if fs.ndim != 2 or fs.shape[0] != nt:
                raise ValueError("Forbidden states array is an incorrect size. Make sure guard levels are accounted for!")   # ArgumentError, :216-219
            W = np.zeros((nt, nt), dtype=np.complex128)
            for k in range(fs.shape[1]):
                W += fw[k] * np.outer(fs[:, k], np.conj(fs[:, k]))               # :222-231
            self.forb_states, self.forb_weights = fs, fw
            self.wmat_real = np.asfortranarray(W.real.copy())
            self.wmat_imag = np.asfortranarray(W.imag.copy())
        else:
            self.forb_states, self.forb_weights = np.zeros((1, 1)), np.zeros(1)  # :233, :237
            self.wmat_real = self.wmat.copy()                                   # :235 (diag)
            self.wmat_imag = np.zeros(nt)                                       # :236 Diagonal(zeros(Ntot))

        self.objFuncType = int(objFuncType)
        self.leak_ubound = float(leak_ubound)

        # memo of the last evaluation (ipopt_interface.jl:27-31, :67-68)
        self.last_leak = 0.0
        self.last_infidelity = 0.0
        self.last_pcof = np.zeros(0)
        self.last_leak_grad = np.zeros(0)
        self.last_infidelity_grad = np.zeros(0)
        self.lastTraceInfidelity = 0.0
        self.lastLeakIntegral = 0.0

        self.saveConvHist = True
        self.objHist = []
        self.primaryHist = []
        self.secondaryHist = []
        self.dualInfidelityHist = []
        self.objThreshold = 0.0
        self.traceInfidelityThreshold = 0.0
        self.usingPriorCoeffs = False
        self.priorCoeffs = np.zeros(0)
        self.quiet = False
        self.save_pcof_hist = False
        self.pcof_hist = []
        self.sv_type = 1
        if Integrator not in (Stormer_Verlet, Implicit_Midpoint):
            raise NotImplementedError("Integrator must be Stormer_Verlet (1) or Implicit_Midpoint (2)")
        self.Integrator_id = Integrator

        if linear_solver is None:
            linear_solver = lsolver_object(nrhs=self.N)                          # :162
        self.linear_solver = linear_solver

    # -- helpers mirroring free functions of the reference that mutate params ------------------
    def estimate_Neumann(self, tol, maxpar):
        """estimate_Neumann!(tol, params, maxpar): src/evalobjgrad.jl:2891-2928."""
        nterms = setup_utils.estimate_Neumann_terms(tol, self.T, self.nsteps, self.Hanti_ops, list(maxpar))
        if nterms > 0:
            self.linear_solver.max_iter = nterms
        return nterms

    def shift_weights_reference(self):
        """Per-level factor of the risk-neutral perturbation of diag(Hconst):
        Hconst[j,j] += ep * 0.01*10^(j-2), j = 2..Ntot (1-based) (src/ipopt_interface.jl:41-44)."""
        s = np.zeros(self.Ntot)
        for j in range(2, self.Ntot + 1):
            s[j - 1] = 0.01 * (10.0 ** (j - 2))
        return s
