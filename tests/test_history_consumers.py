"""State-history consumers: level classification (CPU) and the device-side
population reductions of jq_state_populations against the same reductions done with numpy on the full
history returned by jq_state_history (GPU)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(__file__)


class _P:  # the fields the classifiers read
    def __init__(self, Ne, Ng):
        self.Ne, self.Ng = list(Ne), list(Ng)
        self.Nt = [a + b for a, b in zip(Ne, Ng)]
        self.Nosc = len(Ne)
        self.N = int(np.prod(Ne))
        self.Ntot = int(np.prod(self.Nt))
        self.Nguard = self.Ntot - self.N


def test_level_classification_matches_the_reference_loops():
    import juqbox_jl_amd as jq
    # one oscillator, 3 essential + 2 guard: guard = rows 4,5; forbidden = last row (plotstatectrl.jl:295-297, :343-345)
    p = _P([3], [2])
    assert jq.identify_guard_levels(p).tolist() == [False, False, False, True, True]
    assert jq.identify_forbidden_levels(p).tolist() == [False, False, False, False, True]
    assert jq.identify_guard_levels(p, custom=1).tolist() == [False, True, False, True, False]
    assert not jq.identify_forbidden_levels(_P([2], [0])).any()
    # two oscillators Ne=[2,2], Ng=[1,0]: Nt=[3,2]; k = (q2-1)*3 + q1
    p = _P([2, 2], [1, 0])
    assert jq.identify_guard_levels(p).tolist() == [False, False, True, False, False, True]
    assert jq.identify_forbidden_levels(p).tolist() == [False, False, True, False, False, True]   # only q1 == Nt[1] (Ng[2] == 0)
    # three oscillators (cnot3's layout): brute-force restatement of the triple loops
    p = _P([2, 2, 1], [2, 2, 5])
    guard, forb, lev3 = [], [], []
    for q3 in range(1, p.Nt[2] + 1):
        for q2 in range(1, p.Nt[1] + 1):
            for q1 in range(1, p.Nt[0] + 1):
                guard.append(q1 > p.Ne[0] or q2 > p.Ne[1] or q3 > p.Ne[2])
                forb.append(q1 == p.Nt[0] or q2 == p.Nt[1] or q3 == p.Nt[2])
                lev3.append(q3 == 3)
    assert jq.identify_guard_levels(p).tolist() == guard
    assert jq.identify_forbidden_levels(p).tolist() == forb
    assert jq.specify_level3(p, 2).tolist() == lev3
    # marginalize3 on a synthetic history: probabilities of the third subsystem
    rng = np.random.default_rng(0)
    hist = rng.standard_normal((p.Ntot, p.N, 5)) + 1j * rng.standard_normal((p.Ntot, p.N, 5))
    mp = jq.marginalize3(p, hist)
    assert mp.shape == (p.Nt[2], p.N, 5)
    ref = (np.abs(hist) ** 2).reshape((p.Nt[2], p.Nt[0] * p.Nt[1], p.N, 5)).sum(axis=1)
    assert np.allclose(mp, ref, rtol=1e-14, atol=0)
    assert jq.marginalize3(_P([2, 2], [1, 1]), hist) is None


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["swap02", "cnot2", "cnot3"])
def test_device_reductions_match_numpy_on_the_full_history(jq, name):
    p, info = jq.cases.BUILDERS[name]()
    g = json.load(open(os.path.join(HERE, "golden", "%s.json" % info["golden"])))
    pcof = np.array(g["pcof0"], dtype=float)
    wa = jq.Working_Arrays_HIP(p, pcof.size)
    _, hist, _ = jq.traceobjgrad(pcof, p, wa, True, False)
    P = hist.real ** 2 + hist.imag ** 2                             # [Ntot, N, nsteps+1]
    close = lambda a, b: np.allclose(a, b, rtol=1e-15, atol=1e-300)     # (the device may contract a*a + b*b into an FMA)
    for every in (1, 37):
        pop, maxpop = jq.state_populations(pcof, p, wa, every=every)
        assert pop.shape == (p.Ntot, p.N, p.nsteps // every + 1)
        assert close(pop, P[:, :, ::every][:, :, :pop.shape[2]])
        assert close(maxpop, P.max(axis=(1, 2)))
    if p.Nosc == 3:
        mp = jq.marginalize3_device(pcof, p, wa, every=5)
        ref = jq.marginalize3(p, hist)[:, :, ::5]
        assert mp.shape == ref.shape and np.allclose(mp, ref, rtol=1e-13, atol=1e-300)
    lev, mx, overall = jq.forbidden_level_maxima(pcof, p, wa)
    forb = jq.identify_forbidden_levels(p)
    assert np.array_equal(lev, np.nonzero(forb)[0])
    assert close(mx, P.max(axis=(1, 2))[forb]) and overall == mx.max()
    # a group map with skipped rows
    grp = np.where(jq.identify_guard_levels(p), 1, 0).astype(np.int32)
    grp[0] = -1
    pop2, _ = jq.state_populations(pcof, p, wa, groups=grp, every=101, want_max=False)
    Ps = P[:, :, ::101]
    assert np.allclose(pop2[0], Ps[(grp == 0)].sum(axis=0), rtol=1e-13, atol=1e-300)
    assert np.allclose(pop2[1], Ps[(grp == 1)].sum(axis=0), rtol=1e-13, atol=1e-300)
    # errors
    with pytest.raises(RuntimeError):
        jq.state_populations(pcof, p, wa, every=0)
    wa.close()
