"""d1 x d2 x d3 Kronecker problems with d3 = 7, 8 (Ntot <= 96): native kernels (JQ_EMBED=0) against the embedded twin on the
NT = 7, 8 quad-layout / cooperative-quad kernels.  python scripts/time_embed_big.py"""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import juqbox_jl_amd as jq


def problem(dims, N, nsteps, rng):
    d1, d2, d3 = dims
    Ntot = d1 * d2 * d3

    def op(anti, parts):
        a = np.zeros((Ntot, Ntot))
        if parts & 1:
            for b in range(0, Ntot, d1):
                blk = rng.standard_normal((d1, d1))
                a[b:b + d1, b:b + d1] = blk - blk.T if anti else blk + blk.T
        for stride, bit, period in ((d1, 2, d1 * d2), (d1 * d2, 4, Ntot)):
            if parts & bit:
                for i in range(Ntot - stride):
                    if i // period != (i + stride) // period:
                        continue
                    a[i, i + stride] = rng.standard_normal()
                    a[i + stride, i] = -a[i, i + stride] if anti else a[i, i + stride]
        return a
    Nc = 3
    Hs = [op(False, (7, 2, 4)[q]) for q in range(Nc)]
    Ha = [op(True, (7, 2, 4)[q]) for q in range(Nc)]
    H0 = op(False, 7)
    scale = 2.0 / max(1.0, max(np.abs(np.linalg.eigvalsh(h)).max() for h in Hs + [H0]))
    U0 = np.linalg.qr(rng.standard_normal((Ntot, N)))[0]
    Ut = np.linalg.qr(rng.standard_normal((Ntot, N)) + 1j * rng.standard_normal((Ntot, N)))[0]
    p = jq.objparams([N], [Ntot - N], 1.3 * nsteps / 14, nsteps, Uinit=U0, Utarget=Ut, Cfreq=rng.standard_normal((Nc, 2)), Rfreq=np.zeros(Nc),
                     Hconst=H0 * scale, Hsym_ops=[h * scale for h in Hs], Hanti_ops=[h * scale for h in Ha], objFuncType=1,
                     linear_solver=jq.lsolver_object(max_iter=4))
    p.wmat_real = rng.random(Ntot) * (np.arange(Ntot) >= N)
    return p, 0.3 * rng.standard_normal(2 * Nc * 2 * 4)


for dims in [tuple(int(c) for c in a.split("x")) for a in sys.argv[1:]] or ((3, 4, 7), (3, 3, 8), (4, 3, 8), (2, 2, 8)):
    rng = np.random.default_rng(5)
    p, pcof = problem(dims, 4, 2000, rng)
    line = "%d x %d x %d (Ntot %3d):" % (dims + (p.Ntot,))
    for mode in ("0", "1"):
        os.environ["JQ_EMBED"] = mode
        wa = jq.Working_Arrays_HIP(p, pcof.size)
        os.environ.pop("JQ_EMBED")
        for ns in (1, 256, 3072):
            nodes, weights = np.linspace(-1e-3, 1e-3, ns) if ns > 1 else np.zeros(1), np.full(ns, 1.0 / ns)
            for rep in range(2):
                jq.eval_f_g_grad(pcof, p, wa, nodes, weights, True, shift=np.arange(p.Ntot) * 1e-3)
            t = wa.last_timing()
            line += "  [embed %s] %4d: %.1f ms (fam %d <%d,%d>)" % (mode, ns, t["ms_total"], t["kernel_family"], t["kernel_size"], t["kernel_band"])
        wa.close()
    print(line, flush=True)
