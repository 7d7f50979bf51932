"""Optimiser loop on the GPU path: the reference's callbacks
(eval_f_par / eval_grad_f_par / eval_g_par / eval_jac_g_par, src/ipopt_interface.jl:77-179) driven by
setup_ipopt_problem / run_optimizer.  Ipopt is not in the image; scipy's L-BFGS-B / SLSQP call the same
callbacks, so this checks the glue (memoisation, Tikhonov, history, thresholds, bounds, pcof file), not Ipopt."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)


def _case(jq, name):
    p, info = jq.cases.BUILDERS[name]()
    g = json.load(open(os.path.join(HERE, "golden", "%s.json" % info["golden"])))
    return p, np.array(g["pcof0"], dtype=float), g


def test_lbfgs_descends_from_the_golden_point(jq, tmp_path):
    p, pcof0, g = _case(jq, "swap02")
    p.quiet = True
    wa = jq.Working_Arrays_HIP(p, pcof0.size)
    bound = 4.0 * np.max(np.abs(pcof0))
    prob = jq.setup_ipopt_problem(p, wa, pcof0.size, -bound * np.ones(pcof0.size), bound * np.ones(pcof0.size),
                                  maxIter=12, lbfgsMax=5)
    f0 = prob.eval_f(pcof0)
    assert abs(f0 - np.ravel(g["obj0"])[0]) <= 1e-10 * abs(f0)          # the callbacks see the golden objective
    keep = pcof0.copy()
    base = str(tmp_path / "swap02-opt")
    x = jq.run_optimizer(prob, pcof0, base)
    assert np.array_equal(pcof0, keep)                                    # pcof0 is not overwritten (:423)
    assert prob.obj_val < f0 - 2e-4 and prob.n_iter >= 1
    assert np.all(x >= -bound) and np.all(x <= bound)
    assert len(p.objHist) == prob.n_iter == len(p.primaryHist) == len(p.secondaryHist)
    assert all(b <= a + 1e-12 for a, b in zip(p.objHist, p.objHist[1:]))  # monotone accepted iterates
    assert np.array_equal(jq.read_pcof(base + ".jld2"), x)
    # the optimum found on the GPU is a genuine improvement for the CPU oracle too
    from oracle.oracle import Oracle
    r = Oracle(p).traceobjgrad(x)
    tik = jq.setup_utils.tikhonov_pen(x, p.tik0, None)
    assert abs(r["objfv"] + tik - prob.obj_val) <= 1e-9
    wa.close()


def test_threshold_stops_the_loop(jq):
    p, pcof0, _ = _case(jq, "swap02")
    p.quiet = True
    wa = jq.Working_Arrays_HIP(p, pcof0.size)
    p.objThreshold = 10.0                     # any iterate is "good enough"
    prob = jq.setup_ipopt_problem(p, wa, pcof0.size, -np.ones(pcof0.size), np.ones(pcof0.size), maxIter=20)
    jq.run_optimizer(prob, pcof0)
    assert prob.n_iter == 1 and "threshold" in prob.status
    wa.close()


def test_leakage_as_inequality_constraint(jq):
    p, pcof0, g = _case(jq, "cnot2-leakieq")
    p.quiet = True
    assert p.objFuncType == 3
    wa = jq.Working_Arrays_HIP(p, pcof0.size)
    bound = 4.0 * np.max(np.abs(pcof0))
    prob = jq.setup_ipopt_problem(p, wa, pcof0.size, -bound * np.ones(pcof0.size), bound * np.ones(pcof0.size), maxIter=3)
    assert prob.m == 1 and prob.g_U[0] == p.leak_ubound
    f0, g0 = prob.eval_f(pcof0), prob.eval_g(pcof0)[0]
    assert abs(f0 - np.ravel(g["obj0"])[0]) <= 1e-10 * abs(f0) and abs(g0 - np.ravel(g["obj0"])[1]) <= 1e-10 * abs(g0)
    x = jq.run_optimizer(prob, pcof0)
    assert prob.obj_val <= f0 + 1e-12
    assert np.all(np.abs(x) <= bound + 1e-15)
    wa.close()
