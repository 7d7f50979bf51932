"""All-cores CPU baseline of bench.py (TEST INFRASTRUCTURE, like everything under oracle/): `--procs P` worker
processes evaluate one cnot3 traceobjgrad each with the C oracle, concurrently -- independent ensemble samples are the
only parallelism the reference's serial path offers (src/ipopt_interface.jl:38-65).  Prints one JSON line:
{"procs": P, "seconds": wall time of the P concurrent evaluations, "evals_per_s": P / seconds}.
Runs no GPU code and reads nothing outside the repository."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_ORC = None
_PCOF = None


def _init():
    global _ORC, _PCOF
    import numpy as np
    import juqbox_jl_amd as jq
    from oracle.oracle import Oracle
    params, _ = jq.cases.cnot3()
    _PCOF = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "cnot3.json")))["pcof0"])
    _ORC = Oracle(params)


def _ready(_):
    return os.getpid()


def _one(_):
    r = _ORC.traceobjgrad(_PCOF)
    return float(r["objfv"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=len(os.sched_getaffinity(0)))
    a = ap.parse_args()
    from oracle.oracle import build
    build()                                        # compile once, before the workers race for it
    with mp.get_context("fork").Pool(a.procs, initializer=_init) as pool:
        pool.map(_ready, range(a.procs), chunksize=1)          # every worker has built its problem
        t0 = time.perf_counter()
        vals = pool.map(_one, range(a.procs), chunksize=1)
        el = time.perf_counter() - t0
    assert max(vals) - min(vals) < 1e-12
    print(json.dumps({"procs": a.procs, "seconds": el, "evals_per_s": a.procs / el, "objfv": vals[0]}))


if __name__ == "__main__":
    main()
