"""Consumers of the state history: the level classifications and population
reductions the reference's verbose branch and plot_results derive from usaver/usavei
(src/plotstatectrl.jl:289-394, :405-423; src/evalobjgrad.jl:1004-1018) -- here computed on the device by
jq_state_populations, so the [Ntot x N x (nsteps+1)] history (199 MB at cnot3) never crosses PCIe."""
import ctypes

import numpy as np

from . import _lib
from .evalobjgrad import _f64, _ptr


def _subsystem_indices(params):
    """0-based (q1, q2, q3) of every row, first subsystem fastest (the loops at plotstatectrl.jl:302-321)."""
    Nt = list(params.Nt) + [1] * (3 - params.Nosc)
    k = np.arange(params.Ntot)
    return k % Nt[0], (k // Nt[0]) % Nt[1], k // (Nt[0] * Nt[1])


def identify_guard_levels(params, custom=0):
    """src/plotstatectrl.jl:289-323"""
    Ntot = params.N + params.Nguard
    guard = np.zeros(Ntot, dtype=bool)
    if params.Nosc == 1:
        if custom == 0:
            guard[params.N:] = True
        else:                       # special case for stirap pulses (1-based rows 2 and 4)
            guard[[1, 3]] = True
    elif params.Nosc in (2, 3):
        q = _subsystem_indices(params)
        for j in range(params.Nosc):
            guard |= q[j] >= params.Ne[j]
    return guard


def identify_forbidden_levels(params, custom=0):
    """src/plotstatectrl.jl:334-374: levels with the highest energy level in at least one subsystem."""
    Ntot = params.N + params.Nguard
    forb = np.zeros(Ntot, dtype=bool)
    if params.Nosc == 1:
        # the reference's test is `custom != 0 & Ntot >= 4`, which Julia parses as custom != (0 & Ntot) >= 4,
        # i.e. a chained comparison that is false for every custom in {0, 1, 2, 3}: only the Ng branch can fire
        if custom != (0 & Ntot) and (0 & Ntot) >= 4:
            forb[[1, 3]] = True
        elif params.Ng[0] > 0:
            forb[Ntot - 1] = True
    elif params.Nosc in (2, 3):
        q = _subsystem_indices(params)
        for j in range(params.Nosc):
            if params.Ng[j] > 0:
                forb |= q[j] == params.Nt[j] - 1
    return forb


def specify_level3(params, Nl3):
    """src/plotstatectrl.jl:377-394 (Nl3 is 0-based)."""
    lev = np.zeros(params.N + params.Nguard, dtype=bool)
    if params.Nosc == 3:
        lev |= _subsystem_indices(params)[2] == Nl3
    return lev


def marginalize3(params, unitaryhist):
    """src/plotstatectrl.jl:405-423 on a host-resident history [Ntot, N, nsteps+1] (complex)."""
    if params.Nosc != 3:
        return None
    q3 = _subsystem_indices(params)[2]
    out = np.zeros((params.Nt[2], params.N, unitaryhist.shape[2]))
    p = np.abs(unitaryhist) ** 2
    for r in range(params.Ntot):
        out[q3[r]] += p[r]
    return out


def state_populations(pcof, params, wa, groups=None, every=1, want_max=True):
    """Device-side reductions of the state history of traceobjgrad(pcof, params, wa, verbose=true).

    groups : None -> one group per level (the population curves of plotunitary), or an int array [Ntot] with
             the group of every row (negative = skip).
    every  : keep steps 0, every, 2*every, ... (plot_results plots a down-sampled history).
    Returns (pop [ngroups, N, nout], maxpop [Ntot] or None).
    """
    L, h = _lib.load(), wa.handle
    pcof = _f64(pcof)
    wa.sync_params()
    Ntot = params.Ntot
    if groups is None:
        grp, ngroups, gptr = None, Ntot, None
    else:
        grp = np.ascontiguousarray(np.asarray(groups, dtype=np.int32))
        if grp.size != Ntot:
            raise ValueError("groups must have one entry per level (Ntot)")
        ngroups = int(grp.max()) + 1
        gptr = grp.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    every = int(every)
    nout = params.nsteps // every + 1 if every >= 1 else 0
    pop = np.zeros((ngroups, params.N, max(nout, 0)), order="F")
    maxpop = np.zeros(Ntot) if want_max else None
    _lib.check(L.jq_state_populations(h, _ptr(pcof), pcof.size, gptr, ngroups, every, nout, _ptr(pop),
                                      _ptr(maxpop) if want_max else None), h)
    return pop, maxpop


def marginalize3_device(pcof, params, wa, every=1):
    """marginalize3(params, unitaryhistory) without materialising the history on the host."""
    if params.Nosc != 3:
        return None
    pop, _ = state_populations(pcof, params, wa, groups=_subsystem_indices(params)[2], every=every, want_max=False)
    return pop


def forbidden_level_maxima(pcof, params, wa, custom=0):
    """The verbose branch's report (src/evalobjgrad.jl:1004-1018): per forbidden level the maximum population
    over all columns and time steps; returns (levels (0-based), maxima, overall maximum)."""
    forb = identify_forbidden_levels(params, custom)
    if params.Ntot <= params.N or not forb.any():
        return np.zeros(0, dtype=int), np.zeros(0), 0.0
    L, h = _lib.load(), wa.handle
    pcof = _f64(pcof)
    wa.sync_params()
    maxpop = np.zeros(params.Ntot)
    _lib.check(L.jq_state_populations(h, _ptr(pcof), pcof.size, None, params.Ntot, 1, params.nsteps + 1, None, _ptr(maxpop)), h)
    lev = np.nonzero(forb)[0]
    return lev, maxpop[lev], float(maxpop[lev].max())
