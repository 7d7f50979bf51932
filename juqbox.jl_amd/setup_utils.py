"""Host-side set-up utilities: the O(1)-cost helpers the reference's set-up scripts call before the
hot path (weights, initial conditions, rotating-frame phases, time-step and Neumann estimates,
Tikhonov terms).  numpy only -- nothing here touches the GPU.

Names and argument meaning mirror the reference (file:line relative to /root/reference):
  wmatsetup           src/evalobjgrad.jl:1544-1669
  orig_wmatsetup      src/evalobjgrad.jl:1683-1808
  setup_rotmatrices   src/evalobjgrad.jl:1822-1886
  initial_cond        src/evalobjgrad.jl:3155-3203
  calculate_timestep  src/evalobjgrad.jl:2944-2965, 2983-3022
  estimate_Neumann    src/evalobjgrad.jl:2891-2928   (returns nterms; see objparams.estimate_Neumann)
  tikhonov_pen/grad   src/evalobjgrad.jl:2291-2351
All return plain numpy arrays; diagonal weight matrices are returned as their diagonal (length Ntot).
"""
import math

import numpy as np


def _wmat_diag(Ne, Ng, orig):
    Ne = [int(x) for x in Ne]
    Ng = [int(x) for x in Ng]
    Nt = [a + b for a, b in zip(Ne, Ng)]
    Ndim = len(Ne)
    assert Ndim in (1, 2, 3)
    Ntot = int(np.prod(Nt))
    w = np.zeros(Ntot)
    coeff = 1.0
    if sum(Ng) > 0:
        if Ndim == 1:
            fact = 0.1
            for q in range(Ng[0]):
                w[Ntot - 1 - q] = fact ** q
            coeff = 1.0
        elif Ndim == 2:
            fact = 1e-3
            nForb = 0
            q = 0
            for i2 in range(1, Nt[1] + 1):
                for i1 in range(1, Nt[0] + 1):
                    if not (i1 <= Ne[0] and i2 <= Ne[1]):
                        t1 = fact ** (Nt[0] - i1) if i1 > Ne[0] else 0.0
                        t2 = fact ** (Nt[1] - i2) if i2 > Ne[1] else 0.0
                        if i1 == Nt[0] or i2 == Nt[1]:
                            nForb += 1
                        w[q] = max(t1, t2)
                    q += 1
            # wmatsetup normalises by 1/nForb (:1608), orig_wmatsetup by 10/nForb (:1747)
            coeff = (10.0 if orig else 1.0) / nForb
        else:
            fact = 1e-3
            nForb = 0
            q = 0
            for i3 in range(1, Nt[2] + 1):
                for i2 in range(1, Nt[1] + 1):
                    for i1 in range(1, Nt[0] + 1):
                        if not (i1 <= Ne[0] and i2 <= Ne[1] and i3 <= Ne[2]):
                            t1 = fact ** (Nt[0] - i1) if i1 > Ne[0] else 0.0
                            t2 = fact ** (Nt[1] - i2) if i2 > Ne[1] else 0.0
                            t3 = fact ** (Nt[2] - i3) if i3 > Ne[2] else 0.0
                            forbFact = 1.0
                            # only orig_wmatsetup keeps this ad hoc factor (:1785-1787)
                            if orig and i3 == Nt[2] and i1 <= Ne[0] and i2 <= Ne[1]:
                                forbFact = 100.0
                            w[q] = forbFact * max(t1, t2, t3)
                            if i1 == Nt[0] or i2 == Nt[1] or i3 == Nt[2]:
                                nForb += 1
                        q += 1
            coeff = 10.0 / nForb  # both variants (:1662, :1801)
    return coeff * w


def wmatsetup(Ne, Ng):
    """diag of the default leakage weight matrix W (src/evalobjgrad.jl:1544-1669)."""
    return _wmat_diag(Ne, Ng, orig=False)


def orig_wmatsetup(Ne, Ng):
    """diag of the alternative weight matrix the test set-ups install (src/evalobjgrad.jl:1683-1808)."""
    return _wmat_diag(Ne, Ng, orig=True)


def setup_rotmatrices(Ne, Ng, fund_freq):
    """Rotating-frame angular frequencies per basis state (src/evalobjgrad.jl:1822-1886).
    Returns a tuple of length Nosc (a single array for Nosc == 1, like the reference)."""
    Nt = [int(a) + int(b) for a, b in zip(Ne, Ng)]
    Nosc = len(Nt)
    assert 1 <= Nosc <= 3
    if Nosc == 1:
        return 2 * np.pi * fund_freq[0] * np.arange(Nt[0], dtype=np.float64)
    eye = [np.ones(n) for n in Nt]
    num = [np.arange(n, dtype=np.float64) for n in Nt]
    if Nosc == 2:
        wa = np.kron(eye[1], num[0])
        wb = np.kron(num[1], eye[0])
        return 2 * np.pi * fund_freq[0] * wa, 2 * np.pi * fund_freq[1] * wb
    w1 = np.kron(eye[2], np.kron(eye[1], num[0]))
    w2 = np.kron(eye[2], np.kron(num[1], eye[0]))
    w3 = np.kron(num[2], np.kron(eye[1], eye[0]))
    return (2 * np.pi * fund_freq[0] * w1, 2 * np.pi * fund_freq[1] * w2, 2 * np.pi * fund_freq[2] * w3)


def initial_cond(Ne, Ng):
    """Canonical unit vectors spanning the essential subspace (src/evalobjgrad.jl:3155-3203)."""
    Ne = [int(x) for x in Ne]
    Ng = [int(x) for x in Ng]
    Nt = [a + b for a, b in zip(Ne, Ng)]
    Ntot = int(np.prod(Nt))
    N = int(np.prod(Ne))
    Ident = np.eye(Ntot)
    U0 = Ident[:, :N].copy()
    if len(Nt) in (2, 3) and sum(Ng) > 0:
        col = 0
        m = 0
        Nt3 = Nt + [1] * (3 - len(Nt))
        Ne3 = Ne + [1] * (3 - len(Ne))
        for k3 in range(1, Nt3[2] + 1):
            for k2 in range(1, Nt3[1] + 1):
                for k1 in range(1, Nt3[0] + 1):
                    guard = (k1 > Ne3[0]) or (k2 > Ne3[1]) or (k3 > Ne3[2])
                    if not guard:
                        U0[:, col] = Ident[:, m]
                        col += 1
                    m += 1
    elif len(Nt) > 3:
        raise NotImplementedError("initial_cond(): length(Nt) = %d is not implemented" % len(Nt))
    return np.asfortranarray(U0)


def calculate_timestep(T, H0, Hsym_ops, Hanti_ops, maxpar, Pmin=40):
    """nsteps = ceil(T * max|eig(H0 + sum maxpar_i (Hsym_i + i Hanti_i))| * Pmin / 2pi)
    (src/evalobjgrad.jl:2944-2965)."""
    K1 = np.array(H0, dtype=np.complex128)
    for i in range(len(Hsym_ops)):
        K1 = K1 + maxpar[i] * np.asarray(Hsym_ops[i]) + 1j * maxpar[i] * np.asarray(Hanti_ops[i])
    lamb = np.linalg.eigvals(K1)
    maxeig = np.max(np.abs(lamb))
    samplerate1 = maxeig * Pmin / (2 * np.pi)
    return int(math.ceil(T * samplerate1))


def estimate_Neumann_terms(tol, T, nsteps, Hanti_ops, maxpar):
    """Number of Neumann terms: ceil(log(tol)/log ||h/2 sum maxpar_j Hanti_j||_2) - 1
    (src/evalobjgrad.jl:2891-2928).  Returns nterms (the caller keeps max_iter when nterms <= 0)."""
    k = float(T) / nsteps
    S = 0.5 * k * maxpar[0] * np.asarray(Hanti_ops[0], dtype=np.float64)
    for j in range(1, len(Hanti_ops)):
        S = S + 0.5 * k * maxpar[j] * np.asarray(Hanti_ops[j], dtype=np.float64)
    normS = np.linalg.norm(S, 2)
    return int(math.ceil(math.log(tol) / math.log(normS))) - 1


def tikhonov_pen(pcof, tik0, prior=None):
    """(tik0 * ||pcof - prior||^2) / Npar  (src/evalobjgrad.jl:2291-2318)."""
    pcof = np.asarray(pcof, dtype=np.float64)
    d = pcof if prior is None else pcof - np.asarray(prior, dtype=np.float64)
    return (tik0 * float(np.dot(d, d))) * (1.0 / pcof.size)


def tikhonov_grad(pcof, tik0, prior=None):
    """2 tik0 (pcof - prior) / Npar  (src/evalobjgrad.jl:2320-2351)."""
    pcof = np.asarray(pcof, dtype=np.float64)
    d = pcof if prior is None else pcof - np.asarray(prior, dtype=np.float64)
    return (2.0 * tik0 * (1.0 / pcof.size)) * d
