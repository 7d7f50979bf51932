#!/usr/bin/env python3
"""profiles/r02_pmc.json (round 1: r01_hbm_traffic.json) from rocprofv3 PMC passes (rocpd sqlite files).

usage: make_traffic_json.py out.json [--version "<jq_version>"] [--samples N] [--nsteps S] <db> [<db> ...]
Every db is one `rocprofv3 --pmc <counters> --kernel-trace` pass of the same command.  FETCH_SIZE / WRITE_SIZE
are in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE under-reports wide (16 B/lane) coalesced reads by
exactly 2x on gfx950, so fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as is.  SQ_* counters are
summed as they are ("sq")."""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"_Z\d+(k_[a-z_]+?)(?:I(.*)E)?(?:v|P|1)", name)
    if not m:
        return name.replace(".kd", "")
    base = m.group(1)
    if m.group(2):
        args = re.findall(r"L[ib](\d+)E", m.group(2))
        if base in ("k_forward", "k_backward"):
            return "%s<%s, %s>" % (base, args[0], args[1])
        return "%s<%s>" % (base, ", ".join(args))
    return base


def main():
    out_path, rest = sys.argv[1], sys.argv[2:]
    version, samples, nsteps = None, None, None
    while rest and rest[0].startswith("--"):
        if rest[0] == "--version":
            version = rest[1]
        elif rest[0] == "--samples":
            samples = int(rest[1])
        elif rest[0] == "--nsteps":      # time steps of ONE sweep of the profiled command (one evaluation, objFuncType 1: one backward sweep)
            nsteps = int(rest[1])
        rest = rest[2:]
    dbs = rest
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(int)
    for path in dbs:
        db = sqlite3.connect(path)
        cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        t = lambda p: [x for x in tabs if x.startswith(p)][0]
        q = ("select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from %s e join %s i on e.pmc_id = i.id "
             "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.kernel_name, i.name"
             % (t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")))
        for k, c, v, n in cur.execute(q):
            acc[short(k)][c] += v
            launches[short(k)] = max(launches[short(k)], n)
    kernels = {}
    for k, cs in acc.items():
        n = max(launches[k], 1)
        e = {"launches": launches[k]}
        if "FETCH_SIZE" in cs:
            e["FETCH_SIZE_KiB_per_launch"] = cs["FETCH_SIZE"] / n
        if "WRITE_SIZE" in cs:
            e["WRITE_SIZE_KiB_per_launch"] = cs["WRITE_SIZE"] / n
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            e["hbm_bytes_per_launch"] = (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0 / n
        sq = {c: v for c, v in cs.items() if c.startswith(("SQ_", "GRBM_"))}
        if sq:
            e["sq"] = sq
            if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "SQ_WAVE_CYCLES" in sq and "k_" in k:
                # one wave per SIMD in the MFMA kernels: MFMA busy cycles (per SIMD) / 4 / wave cycles
                e["mfma_busy_frac_of_wave_cycles"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / 4.0 / sq["SQ_WAVE_CYCLES"]
            # what bench.py's roofline block quotes (per launch): MFMA instructions in units of one v_mfma_f64_16x16x4
            # (2048 FLOP; the quad / JQ_BW_T4 kernels issue v_mfma_f64_4x4x4_4b = 512 FLOP, a quarter), instruction mix, stalls
            if sq.get("SQ_INSTS_MFMA"):
                quarter = k.endswith(", 7>") or k.endswith(", 8>")
                e["mfma_insts_per_launch"] = sq["SQ_INSTS_MFMA"] / n
                e["mfma_flop_per_inst"] = 512 if quarter else 2048
                e["mfma_16x16x4_equiv_per_launch"] = sq["SQ_INSTS_MFMA"] / n / (4.0 if quarter else 1.0)
                if "SQ_INSTS_VALU" in sq:
                    e["valu_per_mfma"] = (sq["SQ_INSTS_VALU"] - sq["SQ_INSTS_MFMA"]) / sq["SQ_INSTS_MFMA"]     # (SQ_INSTS_VALU includes the MFMAs)
            if sq.get("SQ_WAVE_CYCLES"):
                if "SQ_WAIT_ANY" in sq:
                    e["wait_frac"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
                if "SQ_WAIT_INST_ANY" in sq:
                    e["wait_inst_frac"] = sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
        if samples is not None:
            e["samples_per_gpu"] = samples
        if nsteps is not None and launches[k] and k.startswith(("k_forward", "k_backward")):
            # bench.py joins this record only with a run of the same chunking (time steps per propagator launch)
            e["steps_per_launch"] = nsteps / launches[k]
        kernels[k] = e
    import os
    chunk_env = {"JQ_OPTIONS": os.environ.get("JQ_OPTIONS")}      # (options that change the chunking would change steps_per_launch)
    json.dump({"note": __doc__.split("usage:")[1].strip(), "library_version": version, "chunk_env": chunk_env, "kernels": kernels},
              open(out_path, "w"), indent=1)
    print("wrote", out_path, "kernels:", ", ".join(sorted(kernels)))


if __name__ == "__main__":
    main()
