#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) outputs as text for profiles/.

    python tools/rocpd_summary.py stats  <results.db>            # per-kernel time table
    python tools/rocpd_summary.py pmc    <results.db> [filter]   # per-kernel counter averages
    python tools/rocpd_summary.py gaps   <results.db>            # idle time between consecutive kernels (launch-bound paths)
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


def gaps(db):
    """Per kernel: average duration and average idle gap between the end of the previous kernel and its start (gaps over
    50 us -- host pauses between moves -- are left out), then the timeline of 24 consecutive dispatches from the middle."""
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end from kernels order by start"))
    acc = {}
    for i in range(1, len(rows)):
        gap = rows[i][1] - rows[i - 1][2]
        if gap > 50000:
            continue
        a = acc.setdefault(rows[i][0], [0, 0, 0])
        a[0] += 1; a[1] += rows[i][2] - rows[i][1]; a[2] += gap
    print("%-70s %8s %10s %12s" % ("kernel", "calls", "avg_us", "gap_before_us"))
    for name, (n, d, g) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%-70s %8d %10.2f %12.2f" % (name[:70], n, d / n / 1e3, g / n / 1e3))
    tot_d = sum(a[1] for a in acc.values()); tot_g = sum(a[2] for a in acc.values())
    print("busy %.1f ms, idle between kernels %.1f ms (%.1f %%)" % (tot_d / 1e6, tot_g / 1e6, 100.0 * tot_g / max(tot_d + tot_g, 1)))
    m = len(rows) // 2
    t0 = rows[m][1]
    for name, st, en in rows[m:m + 24]:
        print("  +%8.2f us  %6.2f us  %s" % ((st - t0) / 1e3, (en - st) / 1e3, name[:60]))


if __name__ == "__main__":
    if sys.argv[1] == "gaps":
        gaps(sys.argv[2])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
