#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.
usage: rocpd_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = list(cur.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),"
        " max(d.grid_size_x), max(d.workgroup_size_x), max(d.group_segment_size), max(d.private_segment_size)"
        " from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id"
        " group by s.kernel_name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    lines = ["%-70s %6s %14s %14s %14s %14s %6s %8s %6s %8s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct",
                                                                     "grid_x", "wg_x", "lds_B", "scratchB")]
    for r in rows:
        lines.append("%-70s %6d %14.3f %14.3f %14.3f %14.3f %6.2f %8d %6d %8d %8d" % (
            r[0][:70], r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9]))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main()
