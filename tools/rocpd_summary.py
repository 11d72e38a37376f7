#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) outputs as text for profiles/.

    python tools/rocpd_summary.py stats  <results.db>            # per-kernel time table
    python tools/rocpd_summary.py pmc    <results.db> [filter]   # per-kernel counter averages
"""
import sqlite3
import sys


def stats(db):
    c = sqlite3.connect(db)
    q = ("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
         "from kernels group by name order by 3 desc")
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("%-78s %8s %12s %7s %11s %11s %11s" % ("kernel", "calls", "total_ms", "%", "avg_us", "min_us", "max_us"))
    for name, n, s, a, mn, mx in rows:
        print("%-78s %8d %12.3f %7.2f %11.2f %11.2f %11.2f" % (name[:78], n, s / 1e6, 100.0 * s / tot, a / 1e3,
                                                          mn / 1e3, mx / 1e3))


def pmc(db, flt=None):
    c = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
         "group by kernel_name, counter_name order by kernel_name, counter_name")
    print("%-60s %-28s %7s %14s %14s %14s" % ("kernel", "counter", "disp", "avg", "min", "max"))
    for k, cn, n, a, mn, mx in c.execute(q):
        if flt and flt not in k:
            continue
        print("%-60s %-28s %7d %14.5g %14.5g %14.5g" % (k[:60], cn, n, a, mn, mx))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
