#!/opt/conda/bin/python3.9
"""Extract the reference's golden vectors into small JSON/.dat fixtures.

Run HERE (build container) with the interpreter that has h5py:

    /opt/conda/bin/python3.9 tests/golden/make_golden.py

Reads (data only, never source code):
  /root/reference/test/reference_solutions/<case>-ref.jld2   keys obj0, grad0
      (written by test/cases/refSol.jl:1-41 through test/evalGrad.jl:33; the
       JLD2 container is valid HDF5)
  /root/reference/test/reference_solutions/err-mat-ref.jld2  key err_mat
      (test/test-stormer-verlet.jl:161-172)
  /root/reference/test/cases/<case>.dat                      start vectors pcof0
      (read by the setups with readdlm, e.g. test/cases/cnot3-setup.jl:270)
Writes tests/golden/<case>.json {obj0, grad0, pcof0, provenance{...sha256}}.
Values are stored with repr() round-trip precision (17 significant digits).
"""
import hashlib
import json
import os
import sys

import h5py
import numpy as np

REF = "/root/reference/test"
OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # case -> pcof .dat file (None: closed-form pcof, built by the test from the setup constants)
    "rabi": None,                      # test/cases/rabi-setup.jl:151-157 (startFromScratch=true)
    "swap02": "swap02.dat",
    "flux": "flux.dat",
    "cnot2": "cnot2.dat",
    "cnot2-leakieq": "cnot2-leakieq.dat",
    "cnot2-jacobi": "cnot2-jacobi.dat",
    "cnot3": "cnot3.dat",
}


def sha256(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main():
    for case, dat in CASES.items():
        jld = os.path.join(REF, "reference_solutions", case + "-ref.jld2")
        with h5py.File(jld, "r") as h:
            obj0 = np.atleast_1d(np.asarray(h["obj0"][()], dtype=np.float64))
            grad0 = np.asarray(h["grad0"][()], dtype=np.float64)
        rec = {
            "case": case,
            "obj0": [float(x) for x in obj0],
            "grad0": [float(x) for x in grad0],
            "grad0_norm2": float(np.linalg.norm(grad0)),
            "provenance": {
                "golden_file": "test/reference_solutions/%s-ref.jld2" % case,
                "golden_sha256": sha256(jld),
                "tolerance": "rtol 1e-10 / atol 1e-14 (test/evalGrad.jl:4-5)",
            },
        }
        if dat is not None:
            p = os.path.join(REF, "cases", dat)
            pcof0 = np.loadtxt(p).ravel()
            rec["pcof0"] = [float(x) for x in pcof0]
            rec["provenance"]["pcof_file"] = "test/cases/" + dat
            rec["provenance"]["pcof_sha256"] = sha256(p)
        with open(os.path.join(OUT, case + ".json"), "w") as f:
            json.dump(rec, f, indent=0)
        print(case, "obj0", rec["obj0"], "len(grad0)", len(grad0), "|grad0|", rec["grad0_norm2"])

    jld = os.path.join(REF, "reference_solutions", "err-mat-ref.jld2")
    with h5py.File(jld, "r") as h:
        em = np.asarray(h["err_mat"][()], dtype=np.float64)
    # h5py shows Julia's column-major (3,2,4) array as (4,2,3); store in Julia index order [cfl, {g,e}, testcase]
    em_julia = np.transpose(em, (2, 1, 0))
    rec = {
        "err_mat_shape_julia": list(em_julia.shape),
        "err_mat": em_julia.tolist(),
        "provenance": {"golden_file": "test/reference_solutions/err-mat-ref.jld2", "golden_sha256": sha256(jld),
                       "tolerance": "max abs diff <= 1e-13 (test/test-stormer-verlet.jl:172)"},
    }
    with open(os.path.join(OUT, "err-mat.json"), "w") as f:
        json.dump(rec, f, indent=0)
    print("err_mat", em_julia.shape)


if __name__ == "__main__":
    sys.exit(main())
