#!/usr/bin/env python3
"""Extract the reference's IMPLICIT-MIDPOINT golden vectors into JSON fixtures.

Run HERE (build container; reads /root/reference, DATA files only):

    python tests/golden/make_golden_imr.py

Reads test/reference_solutions/<case>-ref-imr.jld2 (keys obj0, grad0; written by the second loop of
test/runtests.jl:60-80 with params.Integrator_id = 2 and lsolver_object(solver=JACOBI_SOLVER_M, max_iter=100,
tol=1e-12)) and err-mat-imr-ref.jld2 (test/test-implicit-midpoint.jl) with the repository's own pure-Python
JLD2 reader (juqbox.jl_amd/pcof_io.py, itself checked against h5py extractions in tests/test_pcof_io.py).
Writes tests/golden/<case>-imr.json {obj0, grad0, provenance}; pcof0 is the one of tests/golden/<case>.json.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import juqbox_jl_amd as jq  # noqa: E402

REF = "/root/reference/test/reference_solutions"
CASES = ["rabi", "swap02", "flux", "cnot2", "cnot2-leakieq", "cnot2-jacobi", "cnot3"]


def sha256(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main():
    for case in CASES:
        f = os.path.join(REF, case + "-ref-imr.jld2")
        d = jq.read_jld2(f)
        rec = {"case": case + "-imr",
               "obj0": [float(x) for x in np.atleast_1d(d["obj0"])],
               "grad0": [float(x) for x in np.ravel(d["grad0"])],
               "solver": {"solver": "JACOBI_SOLVER_M", "max_iter": 100, "tol": 1e-12},
               "provenance": {"golden_file": "test/reference_solutions/%s-ref-imr.jld2" % case, "golden_sha256": sha256(f),
                              "generator": "tests/golden/make_golden_imr.py"}}
        json.dump(rec, open(os.path.join(HERE, case + "-imr.json"), "w"), indent=1)
        print(case, len(rec["grad0"]), rec["obj0"])
    f = os.path.join(REF, "err-mat-imr-ref.jld2")
    em = jq.read_jld2(f)["err_mat"]
    json.dump({"err_mat_shape_julia": list(em.shape), "err_mat": em.tolist(),
               "provenance": {"golden_file": "test/reference_solutions/err-mat-imr-ref.jld2", "golden_sha256": sha256(f)}},
              open(os.path.join(HERE, "err-mat-imr.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
