"""GPU (-m gpu): randomized problems vs the CPU oracle -- sizes, paddings and code paths the reference's
own cases do not reach: dense (non-banded) Hamiltonians, every tile count NT = 1..6, Hilbert dimensions
that are not multiples of 16, N = 1..16 columns per sample (ragged slabs), 1..4 controls, 1..3 carrier
frequencies, Neumann terms 0..7, objFuncType 1/2/3, time loops split into several chunks of odd length
(JQ_CHUNK_STEPS), ensembles that do not fill their last slab / workgroup."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


TOL = 1e-10   # the reference's own tolerance (test/evalGrad.jl:4-5); observed ~1e-13 on these random problems


def random_problem(jq, rng, Ntot, N, Nc, Nfreq, nsteps, m, objFuncType, banded):
    T = 1.0 + rng.random()
    if banded == "t4":   # Kronecker structure one level finer (fastest subsystem 4 levels, next one 4, slowest Ntot/16):
        # dense 4x4 diagonal blocks, DIAGONAL couplings (i, i+-4) inside a 16-row block and (i, i+-16) between blocks
        # -> the JQ_BW_T4 slab kernels (v_mfma_f64_4x4x4 + DPP FMAs).  Controls 2, 3, 4 (if present) act on ONE
        # subsystem each, i.e. touch one part of the image only (the trace products then skip the other parts).
        def kron(anti, parts=7):
            a = np.zeros((Ntot, Ntot))
            if parts & 1:
                for b in range(0, Ntot, 4):
                    e = min(b + 4, Ntot)
                    blk = rng.standard_normal((e - b, e - b))
                    a[b:e, b:e] = blk - blk.T if anti else blk + blk.T
            for d, bit in ((4, 2), (16, 4)):
                if not parts & bit:
                    continue
                for i in range(Ntot - d):
                    if d == 4 and (i // 16 != (i + 4) // 16):
                        continue
                    a[i, i + d] = rng.standard_normal()
                    a[i + d, i] = -a[i, i + d] if anti else a[i, i + d]
            return a
        parts = [7, 1, 2, 4]
        Hs = [kron(False, parts[q % 4]) for q in range(Nc)]
        Ha = [kron(True, parts[q % 4]) for q in range(Nc)]
        H0 = kron(False)
    elif banded == "od":   # Kronecker structure (slowest subsystem x 16 fast levels): dense 16x16 diagonal blocks,
        # DIAGONAL first off-diagonal blocks -> the JQ_BW_OD kernels (MFMA for the diagonal blocks only)
        def kron(anti):
            a = np.zeros((Ntot, Ntot))
            for b in range(0, Ntot, 16):
                e = min(b + 16, Ntot)
                blk = rng.standard_normal((e - b, e - b))
                a[b:e, b:e] = blk - blk.T if anti else blk + blk.T
            for i in range(Ntot - 16):
                a[i, i + 16] = rng.standard_normal()
                a[i + 16, i] = -a[i, i + 16] if anti else a[i, i + 16]
            return a
        Hs = [kron(False) for _ in range(Nc)]
        if Nc > 1:                           # one control of the slowest subsystem only: no diagonal blocks at all
            for b in range(0, Ntot, 16):
                Hs[-1][b:b + 16, b:b + 16] = 0.0
        Ha = [kron(True) for _ in range(Nc)]
        if Nc > 1:
            for b in range(0, Ntot, 16):
                Ha[-1][b:b + 16, b:b + 16] = 0.0
        H0 = kron(False)
    elif banded:      # nearest-level couplings like a ladder operator: block band 0 or 1
        def sym():
            a = np.zeros((Ntot, Ntot))
            for i in range(Ntot - 1):
                a[i, i + 1] = rng.standard_normal()
            return a
        Hs = [(lambda a: a + a.T)(sym()) for _ in range(Nc)]
        Ha = [(lambda a: a - a.T)(sym()) for _ in range(Nc)]
        H0 = np.diag(rng.standard_normal(Ntot))
    else:
        Hs = [(lambda a: a + a.T)(rng.standard_normal((Ntot, Ntot))) for _ in range(Nc)]
        Ha = [(lambda a: a - a.T)(rng.standard_normal((Ntot, Ntot))) for _ in range(Nc)]
        H0 = (lambda a: a + a.T)(rng.standard_normal((Ntot, Ntot)))
    scale = 2.0 / max(1.0, max(np.abs(np.linalg.eigvalsh(h)).max() for h in Hs + [H0]))
    H0 *= scale
    Hs = [h * scale for h in Hs]
    Ha = [h * scale for h in Ha]
    Ne, Ng = [N], [Ntot - N]
    U0 = np.linalg.qr(rng.standard_normal((Ntot, N)))[0]
    Ut = np.linalg.qr(rng.standard_normal((Ntot, N)) + 1j * rng.standard_normal((Ntot, N)))[0]
    Cfreq = rng.standard_normal((Nc, Nfreq))
    p = jq.objparams(Ne, Ng, T, nsteps, Uinit=U0, Utarget=Ut, Cfreq=Cfreq, Rfreq=np.zeros(Nc), Hconst=H0,
                     Hsym_ops=Hs, Hanti_ops=Ha, objFuncType=objFuncType,
                     linear_solver=jq.lsolver_object(max_iter=m))
    p.wmat_real = rng.random(Ntot) * (np.arange(Ntot) >= N)
    D1 = int(rng.integers(3, 7))
    pcof = 0.3 * rng.standard_normal(2 * Nc * Nfreq * D1)
    return p, pcof


CASES = [
    # Ntot, N, Nc, Nfreq, nsteps, m, objFuncType, banded, chunk_steps
    (2, 1, 1, 1, 5, 0, 1, False, 0),
    (3, 3, 1, 2, 17, 2, 1, False, 4),
    (4, 2, 4, 1, 11, 1, 3, False, 0),
    (5, 4, 2, 1, 13, 3, 2, False, 3),
    (7, 2, 2, 1, 33, 3, 3, False, 7),
    (8, 3, 3, 2, 9, 2, 1, False, 0),
    (10, 3, 2, 2, 15, 3, 3, False, 4),
    (12, 4, 2, 2, 14, 5, 1, False, 0),
    (16, 16, 1, 1, 9, 1, 1, False, 0),
    (17, 5, 2, 2, 21, 4, 2, False, 5),
    (30, 4, 3, 1, 12, 3, 1, True, 0),
    (33, 7, 1, 3, 11, 5, 3, True, 3),
    (47, 3, 4, 1, 8, 2, 1, False, 0),
    (50, 8, 2, 1, 7, 7, 1, True, 2),
    (64, 4, 3, 2, 6, 3, 3, False, 0),
    (80, 2, 1, 1, 5, 2, 1, True, 0),
    (81, 9, 2, 1, 5, 1, 2, False, 2),
    (96, 4, 3, 1, 6, 6, 1, False, 4),
    (95, 6, 2, 2, 4, 3, 3, True, 0),
    (32, 4, 1, 1, 9, 2, 1, "od", 0),
    (40, 3, 2, 2, 7, 3, 3, "od", 3),
    (80, 5, 3, 1, 6, 4, 2, "od", 0),
    (96, 4, 3, 2, 5, 6, 1, "od", 2),
    (16, 4, 2, 1, 9, 3, 1, "t4", 0),
    (24, 3, 4, 1, 8, 2, 3, "t4", 3),
    (36, 5, 3, 2, 7, 4, 2, "t4", 0),
    (48, 4, 4, 1, 6, 3, 1, "t4", 2),
    (62, 2, 2, 1, 7, 5, 3, "t4", 0),
    (80, 8, 4, 1, 5, 4, 1, "t4", 0),
    (96, 4, 4, 2, 5, 6, 2, "t4", 2),
    # edge cases of the cooperative-quad kernels (mode auto) and their siblings: no / one Neumann term (publication counts),
    # one column, one control, one / two / three time steps (staging ring shorter than its depth), chunks of one step
    (32, 1, 1, 1, 3, 0, 1, "t4", 0),
    (48, 3, 2, 1, 2, 1, 3, "t4", 1),
    (64, 6, 3, 1, 1, 3, 1, "t4", 0),
    (96, 2, 1, 2, 4, 7, 2, "t4", 3),
    # 4 x 4 x 7 and 4 x 4 x 8 (Ntot 97 .. 128): JQ_BW_T4 slab kernels and quad-layout kernels with NT = 7, 8, cooperative-quad
    # kernels with NT = 7 (no cooperative kernels: "coop" falls back to the slab kernels, "slab-od" to the Ntot > 96 cooperative ones)
    (112, 4, 3, 1, 5, 6, 1, "t4", 2),
    (128, 3, 2, 2, 4, 3, 2, "t4", 0),
    (100, 2, 1, 1, 6, 4, 3, "t4", 3),
]


@pytest.mark.parametrize("mode", ["auto", "quad4", "quad8", "quad12", "coop", "slab", "slab-od", "lane", "nolane"])
@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "Ntot%d_N%d_Nc%d_f%d_m%d_o%d_%s_c%d" % (c[0], c[1], c[2], c[3], c[5], c[6], c[7] if isinstance(c[7], str) else ("band" if c[7] else "dense"), c[8]))
def test_random_problem_matches_oracle(jq, cfg, mode):
    """mode 'auto': Ntot <= 16 runs on the row-lane kernels (one lane per (row, column); small batches), Ntot > 16
    with small batches on the cooperative (row-split) kernels; mode 'slab': JQ_COOP_MAX=0 forces the
    one-wave-per-slab kernels that large ensembles use; mode 'lane': JQ_ROWLANE_MAX=0 forces the lane kernels
    (one lane per column) that large ensembles of small systems use; mode 'nolane': JQ_LANE=0 keeps the MFMA
    kernels covered for Ntot <= 16."""
    from oracle.oracle import Oracle
    Ntot, N, Nc, Nfreq, nsteps, m, oft, banded, chunk = cfg
    if (mode == "lane" and Ntot > 8) or (mode == "nolane" and Ntot > 16):
        pytest.skip("lane kernels only exist for Ntot <= 8, row-lane kernels for Ntot <= 16")
    if mode in ("quad4", "quad8", "quad12") and (banded != "t4" or Ntot <= 16):
        pytest.skip("quad4 / quad8 / quad12: the JQ_BW_T4 problems (auto: cooperative-quad kernels) on the quad-layout kernels with "
                    "one (JQ_CQ=0) / two / three (JQ_QUAD8) slabs per workgroup")
    if mode == "coop" and (banded != "t4" or Ntot <= 16):
        pytest.skip("coop: the JQ_BW_T4 problems (auto: quad-layout kernels) once more on the cooperative kernels (JQ_QUAD=0)")
    if mode == "slab-od" and banded != "t4":
        pytest.skip("slab-od: the JQ_BW_T4 problems once more on the JQ_BW_OD / band kernels (JQ_T4=0)")
    rng = np.random.default_rng(1000 + Ntot * 31 + N)
    p, pcof = random_problem(jq, rng, Ntot, N, Nc, Nfreq, nsteps, m, oft, banded)
    opts = {}
    if chunk:
        opts["chunk_steps"] = chunk
    if mode in ("quad8", "quad12"):
        opts["quad8"] = 1 if mode == "quad8" else 2
    if mode in ("slab", "slab-od"):
        opts["coop_max"] = 0
        opts["lane"] = 0
        opts["dq"] = 0      # (round 6: two tile rows without the structure would take the dense cooperative-quad kernels)
    if mode in ("slab", "slab-od", "coop"):
        opts["quad"] = 0
    if mode == "slab-od":
        opts["t4"] = 0
    if mode == "quad4":
        opts["cq"] = 0
    if mode == "nolane":
        opts["lane"] = 0
    if mode == "lane":
        opts["rowlane_max"] = 0
    wa = jq.Working_Arrays_HIP(p, pcof.size, options=opts)
    r = Oracle(p, use_sparse=False).traceobjgrad(pcof, history=True)
    objfv, tg, prim, sec, tinf, ig, lg = jq.traceobjgrad(pcof, p, wa, False, True)
    gn = np.linalg.norm(r["totalgrad"])
    assert abs(prim - r["primaryobjf"]) <= TOL
    assert abs(sec - r["secondaryobjf"]) <= TOL * max(abs(r["secondaryobjf"]), 1e-3)
    assert np.linalg.norm(tg - r["totalgrad"]) <= TOL * gn
    assert np.linalg.norm(ig - r["infidelgrad"]) <= TOL * gn
    if oft != 1:
        assert np.linalg.norm(lg - r["leakgrad"]) <= TOL * gn
    if banded == "t4" and mode in ("slab", "slab-od"):      # the kernel variant under test really ran
        t = wa.last_timing()
        assert t["kernel_family"] == (0 if Ntot <= 96 or mode == "slab" else 1) and (t["kernel_band"] == 8) == (mode == "slab")
    if mode in ("quad4", "quad8", "quad12"):
        assert wa.last_timing()["kernel_family"] == 6
    if banded == "t4" and Ntot > 16 and mode in ("auto", "coop"):      # auto: the cooperative-quad (latency) kernels
        assert wa.last_timing()["kernel_family"] == ((8 if mode == "auto" else 1) if Ntot <= 96 else ((8 if Ntot <= 112 else 6) if mode == "auto" else 0))
    # per-step states
    _, hist, _ = jq.traceobjgrad(pcof, p, wa, True, False)
    assert np.max(np.abs(hist - r["history"])) < 1e-10
    # ragged ensemble: sample counts that leave slabs / workgroups partly empty
    sps = 16 // N
    # (Ntot <= 8: also more columns than one 64-lane wave of the lane kernels holds)
    for nq in sorted({1, sps + 1, 4 * sps + 1} | ({150 // N} if Ntot <= 16 else set())):
        nodes = 0.1 * rng.standard_normal(nq)
        weights = rng.random(nq)
        shift = rng.standard_normal(Ntot) * 0.05
        shift[0] = 0.0
        ref = Oracle(p, use_sparse=False).eval_f_g_grad(pcof, nodes, weights, shift)
        jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=shift)
        gref = ref["last_infidelity_grad"]
        assert abs(p.last_infidelity - ref["last_infidelity"]) <= TOL * abs(ref["last_infidelity"])
        assert abs(p.last_leak - ref["last_leak"]) <= TOL * max(abs(ref["last_leak"]), 1e-3)
        assert np.linalg.norm(p.last_infidelity_grad - gref) <= TOL * np.linalg.norm(gref)
        if oft != 1:
            assert np.linalg.norm(p.last_leak_grad - ref["last_leak_grad"]) <= TOL * np.linalg.norm(gref)
    wa.close()
