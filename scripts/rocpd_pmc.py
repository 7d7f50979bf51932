#!/usr/bin/env python3
"""Sum rocprofv3 PMC counters (rocpd sqlite) per kernel and counter.  usage: rocpd_pmc.py <results.db> [...]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    tot = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        pmc_t = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
        info_t = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
        disp_t = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
        sym_t = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
        q = ("select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from %s e join %s i on e.pmc_id = i.id "
             "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.kernel_name, i.name"
             % (pmc_t, info_t, disp_t, sym_t))
        for k, c, v, cnt in cur.execute(q):
            tot[k][c] += v
            n[k] = max(n[k], cnt)
    for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
        print("%s  (launches %d)" % (k[:90], n[k]))
        for c in sorted(tot[k]):
            print("    %-36s %.6g" % (c, tot[k][c]))


if __name__ == "__main__":
    main()
