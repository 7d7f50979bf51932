"""Problem set-ups of the reference's test-suite and examples, restated with numpy so that the
hot path can be fed the exact inputs the golden vectors were produced with.

Each builder returns `(params, info)`; `info` carries maxpar, D1, nCoeff and (where the
reference defines one in closed form) `pcof0`.  Start vectors that the reference reads from
`test/cases/*.dat` are data fixtures (tests/golden/*.json), passed in by the caller.

Sources (relative to /root/reference):
  rabi            test/cases/rabi-setup.jl
  swap02          test/cases/swap02-setup.jl
  flux            test/cases/flux-setup.jl
  cnot2(+variants) test/cases/cnot2-setup.jl, cnot2-leakieq-setup.jl, cnot2-jacobi-setup.jl
  cnot3           test/cases/cnot3-setup.jl
  cnot1           examples/cnot1-setup.jl            (Stormer-Verlet selected; own seeded pcof0)
  swap02_rn       examples/Risk_Neutral/swap-02-risk-neutral.jl (+ run_all.jl:67 ep_max)
"""
import math

import numpy as np

from . import setup_utils as su
from .objparams import JACOBI_SOLVER, lsolver_object, objparams

EPS = np.finfo(np.float64).eps


def _lowering(n):
    """Array(Bidiagonal(zeros(n), sqrt.(1:n-1), :U))"""
    a = np.zeros((n, n))
    for i in range(n - 1):
        a[i, i + 1] = math.sqrt(i + 1)
    return a


def _nsteps_from_eig(K1, T, Pmin):
    lamb = np.linalg.eigvals(K1)
    maxeig = np.max(np.abs(lamb))
    samplerate1 = maxeig * Pmin / (2 * np.pi)
    return int(math.ceil(T * samplerate1))


def rabi():
    """test/cases/rabi-setup.jl:47-227 (2-level qubit, X-gate over one Rabi period, no guards)."""
    N, Nguard = 2, 0
    Ntot = N + Nguard
    fa, xa = 0.0, 2 * 0.1099
    T = 2 * np.pi
    theta = np.pi / 2
    aOmega = np.pi / T
    utarget = np.eye(Ntot, N, dtype=np.complex128)
    utarget[0, 0] = math.cos(aOmega * T)
    utarget[1, 0] = -(math.sin(theta) + 1j * math.cos(theta)) * math.sin(aOmega * T)
    utarget[0, 1] = (math.sin(theta) - 1j * math.cos(theta)) * math.sin(aOmega * T)
    utarget[1, 1] = math.cos(aOmega * T)
    omega1 = su.setup_rotmatrices([N], [Nguard], [fa])
    vtarget = np.exp(1j * omega1 * T)[:, None] * utarget
    Nfreq = 1
    om = np.zeros((1, Nfreq))
    number = np.diag(np.arange(Ntot, dtype=np.float64))
    H0 = -0.5 * (2 * np.pi) * xa * (number @ number - number)
    amat = _lowering(Ntot)
    adag = amat.T
    maxpar = 1.0 * aOmega / Nfreq
    K1 = H0 + maxpar * (amat + amat.T) + 1j * maxpar * (amat - amat.T)
    nsteps = _nsteps_from_eig(K1, T, 80)
    U0 = np.eye(Ntot)[:, :N]
    params = objparams([N], [Nguard], T, nsteps, Uinit=U0, Utarget=vtarget, Cfreq=om, Rfreq=[fa],
                       Hconst=H0, Hsym_ops=[amat + adag], Hanti_ops=[amat - adag])
    params.quiet = True
    D1 = 3
    nCoeff = 2 * 1 * Nfreq * D1
    pcof0 = np.zeros(nCoeff)
    pcof0[0:D1] = aOmega * math.cos(theta)
    pcof0[D1:2 * D1] = aOmega * math.sin(theta)
    params.estimate_Neumann(EPS, [maxpar])
    return params, dict(maxpar=[maxpar], D1=D1, nCoeff=nCoeff, pcof0=pcof0, golden="rabi")


def swap02():
    """test/cases/swap02-setup.jl:43-217 (single qudit, 3 essential + 1 guard level, |0>-|2> swap)."""
    N, Nguard = 3, 1
    Ntot = N + Nguard
    T = 150.0
    freq_alice = [0, 4.09947, 3.87409, 3.6206]
    utarget = np.zeros((Ntot, N), dtype=np.complex128)
    utarget[2, 0] = 1
    utarget[1, 1] = 1
    utarget[0, 2] = 1
    omega1 = su.setup_rotmatrices([N], [Nguard], [freq_alice[1]])
    vtarget = np.exp(1j * omega1 * T)[:, None] * utarget
    xa = 2 * 0.1099
    number = np.diag(np.arange(Ntot, dtype=np.float64))
    H0 = -0.5 * (2 * np.pi) * xa * (number @ number - number)
    amat = _lowering(Ntot)
    adag = amat.T
    Nfreq = 2
    om = np.zeros((1, Nfreq))
    om[0, 1] = H0[2, 2]
    maxpar = 2 * np.pi * 0.0132 / Nfreq / 2
    K1 = H0 + maxpar * (amat + amat.T) + 1j * maxpar * (amat - amat.T)
    nsteps = _nsteps_from_eig(K1, T, 80)
    U0 = np.eye(Ntot)[:, :N]
    params = objparams([N], [Nguard], T, nsteps, Uinit=U0, Utarget=vtarget, Cfreq=om,
                       Rfreq=[freq_alice[1]], Hconst=H0, Hsym_ops=[amat + adag], Hanti_ops=[amat - adag])
    params.quiet = True
    params.estimate_Neumann(EPS, [maxpar])
    return params, dict(maxpar=[maxpar], D1=10, nCoeff=40, golden="swap02")


def flux():
    """test/cases/flux-setup.jl:53-222 (4+2 level qudit; second 'control' is a^dag a with a zero
    anti-symmetric partner; tik0 = 0.1; Neumann terms stay at the default 3)."""
    N, Nguard = 4, 2
    Ntot = N + Nguard
    fa, xa = 5.0, 0.2
    T = 11.0
    Ident = np.eye(Ntot)
    utarget = np.eye(Ntot, N, dtype=np.complex128)
    utarget[:, 3] = Ident[:, 2]
    utarget[:, 2] = Ident[:, 3]
    omega1 = su.setup_rotmatrices([N], [Nguard], [fa])
    vtarget = np.exp(1j * omega1 * T)[:, None] * utarget
    Nfreq = 2
    number = np.diag(np.arange(Ntot, dtype=np.float64))
    H0 = -0.5 * (2 * np.pi) * xa * (number @ number - number)
    amat = _lowering(Ntot)
    adag = amat.T
    Hsym_ops = [amat + adag, adag @ amat]
    Hanti_ops = [amat - adag, np.zeros((Ntot, Ntot))]
    Nctrl = 2
    om = np.zeros((Nctrl, Nfreq))
    om[:, 1] = -2.0 * np.pi * xa
    maxpar = 0.08
    max_flux = 2 * np.pi * 5.0
    U0 = Ident[:, :N]
    nsteps = su.calculate_timestep(T, H0, Hsym_ops, Hanti_ops, [maxpar, max_flux])
    params = objparams([N], [Nguard], T, nsteps, Uinit=U0, Utarget=vtarget, Cfreq=om, Rfreq=[fa, fa],
                       Hconst=H0, Hsym_ops=Hsym_ops, Hanti_ops=Hanti_ops, use_sparse=True)
    params.quiet = True
    params.tik0 = 0.1
    params.traceInfidelityThreshold = 1e-5
    return params, dict(maxpar=[maxpar, max_flux], D1=30, nCoeff=240, golden="flux")


def cnot2(variant="cnot2"):
    """test/cases/cnot2-setup.jl:45-267; variant 'cnot2-leakieq' adds objFuncType=3
    (cnot2-leakieq-setup.jl:188), 'cnot2-jacobi' the Jacobi solver (cnot2-jacobi-setup.jl:186)."""
    Ne, Ng = [2, 2], [1, 2]
    Nt1, Nt2 = 3, 4
    Tmax = 100.0
    fa, fb = 4.10595, 4.81526
    x1, x2, x12 = 2 * 0.1099, 2 * 0.1126, 0.1
    a1, a2 = _lowering(Nt1), _lowering(Nt2)
    I1, I2 = np.eye(Nt1), np.eye(Nt2)
    amat = np.kron(I2, a1)
    bmat = np.kron(a2, I1)
    adag, bdag = amat.T, bmat.T
    N1 = np.kron(I2, np.diag(np.arange(Nt1, dtype=np.float64)))
    N2 = np.kron(np.diag(np.arange(Nt2, dtype=np.float64)), I1)
    H0 = -2 * np.pi * (x1 / 2 * (N1 @ N1 - N1) + x2 / 2 * (N2 @ N2 - N2) + x12 * (N1 @ N2))
    amax, bmax = 0.02, 0.05
    maxpar = [amax, bmax]
    K1 = H0 + (amax * (amat + amat.T) + 1j * amax * (amat - amat.T)
               + bmax * (bmat + bmat.T) + 1j * bmax * (bmat - bmat.T))
    nsteps = _nsteps_from_eig(K1, Tmax, 40)
    Hsym_ops = [amat + adag, bmat + bdag]
    Hanti_ops = [amat - adag, bmat - bdag]
    Nfreq = 2
    om = np.zeros((2, Nfreq))
    om[:, 1] = -2.0 * np.pi * x12
    Ntot, N = 12, 4
    utarget = np.zeros((Ntot, N), dtype=np.complex128)
    utarget[0, 0] = 1.0      # Ng1 == 1 branch (:168-172)
    utarget[1, 1] = 1.0
    utarget[3, 3] = 1.0
    utarget[4, 2] = 1.0
    omega1, omega2 = su.setup_rotmatrices(Ne, Ng, [fa, fb])
    vtarget = (np.exp(1j * omega1 * Tmax) * np.exp(1j * omega2 * Tmax))[:, None] * utarget
    U0 = su.initial_cond(Ne, Ng)
    kw = {}
    if variant == "cnot2-leakieq":
        kw = dict(objFuncType=3, leak_ubound=1.0e-3)
    elif variant == "cnot2-jacobi":
        kw = dict(linear_solver=lsolver_object(solver=JACOBI_SOLVER, max_iter=100, tol=1e-15, nrhs=4))
    params = objparams(Ne, Ng, Tmax, nsteps, Uinit=U0, Utarget=vtarget, Cfreq=om, Rfreq=[fa, fb],
                       Hconst=H0, Hsym_ops=Hsym_ops, Hanti_ops=Hanti_ops, use_sparse=False, **kw)
    params.wmat_real = su.orig_wmatsetup(Ne, Ng)
    params.quiet = True
    params.estimate_Neumann(EPS, maxpar)
    return params, dict(maxpar=maxpar, D1=10, nCoeff=80, golden=variant)


def cnot3(Ng3=5, Nfreq=3):
    """test/cases/cnot3-setup.jl:57-318 (qubit-qubit-cavity; Ntot = 4*4*6 = 96, N = 4)."""
    Ne, Ng = [2, 2, 1], [2, 2, Ng3]
    Nt = [a + b for a, b in zip(Ne, Ng)]
    Tmax = 550.0
    fa, fb, fs = 4.10595, 4.81526, 7.8447
    xa = 2 * 0.1099
    xb = 2 * 0.1126
    xs = 0.002494 ** 2 / xa
    xab = 1.0e-6
    xas = math.sqrt(xa * xs)
    xbs = math.sqrt(xb * xs)
    a1, a2, a3 = _lowering(Nt[0]), _lowering(Nt[1]), _lowering(Nt[2])
    I1, I2, I3 = np.eye(Nt[0]), np.eye(Nt[1]), np.eye(Nt[2])
    amat = np.kron(I3, np.kron(I2, a1))
    bmat = np.kron(I3, np.kron(a2, I1))
    cmat = np.kron(a3, np.kron(I2, I1))
    adag, bdag, cdag = amat.T, bmat.T, cmat.T
    num = [np.diag(np.arange(n, dtype=np.float64)) for n in Nt]
    Na = np.kron(I3, np.kron(I2, num[0]))
    Nb = np.kron(I3, np.kron(num[1], I1))
    Nc = np.kron(num[2], np.kron(I2, I1))
    H0 = -2 * np.pi * (xa / 2 * (Na @ Na - Na) + xb / 2 * (Nb @ Nb - Nb) + xs / 2 * (Nc @ Nc - Nc)
                       + xab * (Na @ Nb) + xas * (Na @ Nc) + xbs * (Nb @ Nc))
    amax, bmax, cmax = 0.05, 0.1, 0.1
    maxpar = [amax, bmax, cmax]
    K1 = (H0 + amax * (amat + amat.T) + 1j * amax * (amat - amat.T)
          + bmax * (bmat + bmat.T) + 1j * bmax * (bmat - bmat.T)
          + cmax * (cmat + cmat.T) + 1j * cmax * (cmat - cmat.T))
    nsteps = _nsteps_from_eig(K1, Tmax, 40)
    Hsym_ops = [amat + adag, bmat + bdag, cmat + cdag]
    Hanti_ops = [amat - adag, bmat - bdag, cmat - cdag]
    Ncoupled = 3
    om = np.zeros((Ncoupled, Nfreq))
    if Nfreq == 2:
        om[:, 1] = -1.0 * np.pi * xas
    elif Nfreq == 3:
        om[0:2, 1] = -2.0 * np.pi * xa
        om[0:2, 2] = -2.0 * np.pi * xb
        om[2, 1] = -2.0 * np.pi * xas
        om[2, 2] = -2.0 * np.pi * xbs
    N2tot = Nt[0] * Nt[1]
    N2 = Ne[0] * Ne[1]
    G2 = np.zeros((N2tot, N2), dtype=np.complex128)
    G2[0, 0] = 1.0       # Ng[1] == 2 branch (:201-205)
    G2[1, 1] = 1.0
    G2[4, 3] = 1.0
    G2[5, 2] = 1.0
    I3e = np.eye(Nt[2], Ne[2])
    utarget = np.kron(I3e, G2)
    omega1, omega2, omega3 = su.setup_rotmatrices(Ne, Ng, [fa, fb, fs])
    rot = np.exp(1j * omega1 * Tmax) * np.exp(1j * omega2 * Tmax) * np.exp(1j * omega3 * Tmax)
    vtarget = rot[:, None] * utarget
    U0 = su.initial_cond(Ne, Ng)
    params = objparams(Ne, Ng, Tmax, nsteps, Uinit=U0, Utarget=vtarget, Cfreq=om, Rfreq=[fa, fb, fs],
                       Hconst=H0, Hsym_ops=Hsym_ops, Hanti_ops=Hanti_ops, use_sparse=True)
    params.wmat_real = su.orig_wmatsetup(Ne, Ng)
    params.quiet = True
    params.estimate_Neumann(EPS, maxpar)
    return params, dict(maxpar=maxpar, D1=15, nCoeff=2 * Ncoupled * Nfreq * 15, golden="cnot3")


def cnot1(seed=2456):
    """examples/cnot1-setup.jl:33-133 (single qudit, 4 essential + 2 guard levels), with the
    Stormer-Verlet integrator and the default 3 Neumann terms (the example never calls
    estimate_Neumann!).  The example draws pcof0 = maxpar*0.01*rand(nCoeff) from an unseeded Julia
    RNG (:126,:133); here the same expression uses numpy default_rng(seed) (parity for this
    config is GPU-vs-oracle)."""
    N, Nguard = 4, 2
    Ntot = N + Nguard
    T = 100.0
    fa, xa = 4.10336, 0.2198
    number = np.diag(np.arange(Ntot, dtype=np.float64))
    H0 = -0.5 * (2 * np.pi) * xa * (number @ number - number)
    amat = _lowering(Ntot)
    adag = amat.T
    Hsym_ops = [amat + adag]
    Hanti_ops = [amat - adag]
    maxctrl = 0.001 * 2 * np.pi * 8.5
    nsteps = su.calculate_timestep(T, H0, Hsym_ops, Hanti_ops, [maxctrl])
    Nfreq = 3
    om = np.zeros((1, Nfreq))
    om[0, 1] = -2.0 * np.pi * xa
    om[0, 2] = -2.0 * np.pi * 2.0 * xa
    const_fact = 0.45
    maxamp = np.zeros(Nfreq)
    maxamp[0] = maxctrl * const_fact
    maxamp[1:] = maxctrl * (1.0 - const_fact) / (Nfreq - 1)
    maxpar = float(np.max(maxamp))
    U0 = su.initial_cond([N], [Nguard])
    gate_cnot = np.zeros((N, N), dtype=np.complex128)
    gate_cnot[0, 0] = 1.0
    gate_cnot[1, 1] = 1.0
    gate_cnot[2, 3] = 1.0
    gate_cnot[3, 2] = 1.0
    utarget = U0 @ gate_cnot
    omega1 = su.setup_rotmatrices([N], [Nguard], [fa])
    vtarget = np.exp(1j * omega1 * T)[:, None] * utarget
    params = objparams([N], [Nguard], T, nsteps, Uinit=U0, Utarget=vtarget, Cfreq=om, Rfreq=[fa],
                       Hconst=H0, Hsym_ops=Hsym_ops, Hanti_ops=Hanti_ops)
    params.quiet = True
    D1 = 10
    nCoeff = 2 * 1 * Nfreq * D1
    pcof0 = maxpar * 0.01 * np.random.default_rng(seed).random(nCoeff)
    return params, dict(maxpar=[maxpar], D1=D1, nCoeff=nCoeff, pcof0=pcof0, golden=None)


def swap02_rn(nquad=512, seed=2456, ep_max=2 * np.pi * 2e-2):
    """examples/Risk_Neutral/swap-02-risk-neutral.jl:39-170 (T=300, Nfreq=2, D1=12, target NOT
    rotated: vtarget = utarget, :72) with the Gauss-Legendre ensemble it builds at :45-49
    (nodes*0.5*ep_max, weights*0.5) for ep_max = 2pi*2e-2 (run_all.jl:67).
    pcof0 = (rand(nCoeff)-0.5)*maxpar*0.1 (:151) from numpy default_rng(seed) instead of Julia's RNG;
    quadrature nodes from numpy leggauss instead of FastGaussQuadrature (parity unpinned, section 8c)."""
    N, Nguard = 3, 1
    Ntot = N + Nguard
    T = 300.0
    fa, xa = 4.10336, 0.2198
    number = np.diag(np.arange(Ntot, dtype=np.float64))
    H0 = -0.5 * (2 * np.pi) * xa * (number @ number - number)
    utarget = np.zeros((Ntot, N), dtype=np.complex128)
    utarget[2, 0] = 1
    utarget[1, 1] = 1
    utarget[0, 2] = 1
    vtarget = utarget
    amat = _lowering(Ntot)
    adag = amat.T
    Hsym_ops = [amat + adag]
    Hanti_ops = [amat - adag]
    Nfreq = 2
    om = np.zeros((1, Nfreq))
    om[0, 1] = -2.0 * np.pi * xa
    maxctrl = 2 * np.pi * 1.2e-2
    maxamp = np.full(Nfreq, maxctrl / Nfreq)
    maxpar = float(np.max(maxamp))
    nsteps = su.calculate_timestep(T, H0, Hsym_ops, Hanti_ops, [maxctrl])
    U0 = su.initial_cond([N], [Nguard])
    params = objparams([N], [Nguard], T, nsteps, Uinit=U0, Utarget=vtarget, Cfreq=om, Rfreq=[fa],
                       Hconst=H0, Hsym_ops=Hsym_ops, Hanti_ops=Hanti_ops, wmatScale=1.0)
    params.quiet = True
    D1 = 12
    nCoeff = 2 * 1 * Nfreq * D1
    pcof0 = (np.random.default_rng(seed).random(nCoeff) - 0.5) * maxpar * 0.1
    params.estimate_Neumann(EPS, [maxpar])
    x, w = np.polynomial.legendre.leggauss(nquad)
    nodes = x * 0.5 * ep_max
    weights = w * 0.5
    return params, dict(maxpar=[maxpar], D1=D1, nCoeff=nCoeff, pcof0=pcof0, nodes=nodes, weights=weights,
                        golden=None)


def cnot2_lab(Pmin=200, pcof_file=None):
    """examples/cnot2-lab.jl: LAB-frame evaluation of a CNOT pulse on two coupled qubits (Ne = [2,2], Ng = [1,1], Ntot = 9,
    N = 4, T = 50, Nfreq = 2) with UNCOUPLED controls Hunc_ops = [a + a', b + b'] (:124) -- the forward-only branch of KS!
    (src/evalobjgrad.jl:2373-2387).  H0 keeps the qubit frequencies (:121), Rfreq = rot_freq (:73); the time step resolves
    them: nsteps = ceil(T max|eig(H0 + sum maxpar Hunc)| Pmin / 2 pi) (:133, Pmin = 200).  The example starts from
    drives/cnot2-pcof-opt-t50.jld2 (:208; tests/golden/jld2 holds that data file) -- pass its path as pcof_file; default:
    own seeded coefficients."""
    Ne, Ng = [2, 2], [1, 1]
    Nt1, Nt2 = 3, 3
    Tmax = 50.0
    fa, fb = 4.10595, 4.81526
    x1, x2, x12 = 2 * 0.1099, 2 * 0.1126, 0.1
    a1, a2 = _lowering(Nt1), _lowering(Nt2)
    I1, I2 = np.eye(Nt1), np.eye(Nt2)
    amat = np.kron(I2, a1)
    bmat = np.kron(a2, I1)
    N1 = np.kron(I2, np.diag(np.arange(Nt1, dtype=np.float64)))
    N2 = np.kron(np.diag(np.arange(Nt2, dtype=np.float64)), I1)
    H0 = 2 * np.pi * (fa * N1 + fb * N2 - x1 / 2 * (N1 @ N1 - N1) - x2 / 2 * (N2 @ N2 - N2) - x12 * (N1 @ N2))
    Hunc_ops = [amat + amat.T, bmat + bmat.T]
    maxpar = [0.014, 0.020]
    K1 = H0 + maxpar[0] * Hunc_ops[0] + maxpar[1] * Hunc_ops[1]
    nsteps = _nsteps_from_eig(K1, Tmax, Pmin)
    Nfreq = 2
    om = np.zeros((2, Nfreq))
    om[:, 1] = -2.0 * np.pi * x12
    Ntot, N = 9, 4
    utarget = np.zeros((Ntot, N), dtype=np.complex128)
    utarget[0, 0] = 1.0      # Ng1 == 1 branch (:163-167)
    utarget[1, 1] = 1.0
    utarget[3, 3] = 1.0
    utarget[4, 2] = 1.0
    U0 = su.initial_cond(Ne, Ng)
    params = objparams(Ne, Ng, Tmax, nsteps, Uinit=U0, Utarget=utarget, Cfreq=om, Rfreq=[fa, fb], Hconst=H0,
                       Hunc_ops=Hunc_ops, use_sparse=False)
    params.quiet = True
    if pcof_file is not None:
        from .pcof_io import read_pcof
        pcof0 = read_pcof(pcof_file)
    else:
        pcof0 = (np.random.default_rng(2456).random(2 * 2 * Nfreq * 10) - 0.5) * 0.02
    return params, dict(maxpar=maxpar, D1=pcof0.size // (2 * 2 * Nfreq), nCoeff=pcof0.size, pcof0=pcof0, golden=None,
                        rot_freq=[fa, fb])


def cnot3_dense(eps=1.0e-2, seed=11):
    """cnot3's dimensions (Ntot = 96, N = 4), controls, target, weights and time stepping with a DENSE Hermitian drift:
    Hconst + eps (D + D') with D ~ N(0, 1 / Ntot) -- a user Hamiltonian without Kronecker structure (all-to-all couplings, another
    basis).  Not a case of the reference; it is what north_star calls "the dense (H x state-batch) contraction": the planner finds
    no structure to exploit and the propagators run on dense 16 x 16 x 4 MFMA tiles (kernels <6, 5>; bench.py's `dense_operator`
    block).  The perturbation is small against the drift (|Hconst| ~ 10), so cnot3's nsteps still resolves it."""
    params, info = cnot3()
    rng = np.random.default_rng(seed)
    D = rng.standard_normal((params.Ntot, params.Ntot)) / math.sqrt(params.Ntot)
    params.Hconst = np.asfortranarray(params.Hconst + eps * (D + D.T))
    params.use_sparse = False
    return params, dict(info, golden=None)


BUILDERS = {
    "rabi": rabi,
    "swap02": swap02,
    "flux": flux,
    "cnot2": lambda: cnot2("cnot2"),
    "cnot2-leakieq": lambda: cnot2("cnot2-leakieq"),
    "cnot2-jacobi": lambda: cnot2("cnot2-jacobi"),
    "cnot3": cnot3,
    "cnot1": cnot1,
    "swap02_rn": swap02_rn,
    "cnot2_lab": lambda: cnot2_lab(Pmin=40),
}


def cnot3_ensemble(nsamples, ep_max=2 * np.pi * 1.0e-4):
    """Risk-neutral ensemble for the cnot3 configuration (BASELINE.json configs[3]/[4] combined: the
    reference's own perturbation 0.01*ep*10^(j-2) (src/ipopt_interface.jl:41-44) overflows any
    sensible scale at Ntot = 96, so the ensemble perturbs the drift Hamiltonian with a common
    detuning of the three oscillators instead:  Hconst + ep*(Na + Nb + Nc), ep = Gauss-Legendre
    nodes on [-ep_max, ep_max] (ep_max = 2pi x 100 kHz), weights summing to one -- the same
    construction as examples/Risk_Neutral/swap-02-risk-neutral.jl:45-49.
    Returns (nodes, weights, shift[Ntot])."""
    Nt = [4, 4, 6]
    i1 = np.tile(np.arange(Nt[0]), Nt[1] * Nt[2])
    i2 = np.tile(np.repeat(np.arange(Nt[1]), Nt[0]), Nt[2])
    i3 = np.repeat(np.arange(Nt[2]), Nt[0] * Nt[1])
    shift = (i1 + i2 + i3).astype(np.float64)
    if nsamples <= 4096:
        x, w = np.polynomial.legendre.leggauss(nsamples)
    else:
        # numpy's leggauss diagonalises an n x n companion matrix: O(n^3), eight minutes at n = 24 576 (bench.py's strong-scaling
        # ensemble).  Large ensembles use a COMPOSITE Gauss-Legendre rule instead: k equal panels with n / k <= 4096 nodes each
        # (24 576 = 8 panels of 3 072); sample counts without such a divisor fall back to scipy's O(n) asymptotic rule.
        k = next((k for k in range(2, 65) if nsamples % k == 0 and nsamples // k <= 4096), 0)
        if k:
            xp, wp = np.polynomial.legendre.leggauss(nsamples // k)
            centers = -1.0 + (2.0 * np.arange(k) + 1.0) / k
            x = (centers[:, None] + xp[None, :] / k).ravel()
            w = np.tile(wp / k, k)
        else:
            from scipy.special import roots_legendre
            x, w = roots_legendre(nsamples)
    return x * ep_max, w * 0.5, shift
