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


def cpu_quota():
    """CPU cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota (containers)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def run_level(procs, limit):
    """`procs` concurrent evaluations (one per worker process); None if they do not finish within `limit` seconds."""
    pool = mp.get_context("fork").Pool(procs, initializer=_init)
    try:
        pool.map_async(_ready, range(procs), chunksize=1).get(limit + 60)          # every worker has built its problem
        t0 = time.perf_counter()
        vals = pool.map_async(_one, range(procs), chunksize=1).get(limit)
        el = time.perf_counter() - t0
        assert max(vals) - min(vals) < 1e-12
        return {"procs": procs, "seconds": el, "evals_per_s": procs / el, "objfv": vals[0]}
    except mp.TimeoutError:
        return None
    finally:
        pool.terminate()
        pool.join()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=0, help="0: probe 4, 16, 64, ... up to the CPU quota while it still pays")
    ap.add_argument("--seconds-per-eval", type=float, default=0.0, help="single-core time of one evaluation (sets the time limits)")
    a = ap.parse_args()
    from oracle.oracle import build
    build()                                        # compile once, before the workers race for it
    t1 = a.seconds_per_eval
    if t1 <= 0.0:
        _init()
        t0 = time.perf_counter()
        _one(0)
        t1 = time.perf_counter() - t0
    quota = cpu_quota()
    if a.procs > 0:
        best = run_level(a.procs, 30.0 * t1 + 30.0)
    else:
        # affinity masks of containers overstate the usable cores: grow the process count only while every evaluation still
        # runs at (nearly) single-core speed, bounded in time
        best, p = None, min(4, quota)
        while True:
            r = run_level(p, 3.0 * t1 + 5.0)
            if r is None or (best is not None and r["evals_per_s"] < 1.1 * best["evals_per_s"]):
                break
            best = r
            if p >= quota:
                break
            p = min(4 * p, quota)
    if best is None:
        best = {"procs": 0, "seconds": 0.0, "evals_per_s": 0.0, "error": "no level finished inside its time limit"}
    best["cpu_quota"] = quota
    print(json.dumps(best))


if __name__ == "__main__":
    main()
