#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace and/or PMC counters) as text.

    python tools/rocpd_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(
        "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
        "group by %s order by sum(end-start) desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# kernel-trace summary of %s (durations in us)" % path)
    print("%-58s %7s %12s %12s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, calls, tot, avg, mn, mx in rows:
        short = name.split("(")[0][-58:]
        print("%-58s %7d %12.1f %12.2f %10.2f %10.2f %6.2f" % (short, calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                               100.0 * tot / total))
    try:
        pmc = cur.execute(
            "select k.%s, p.counter_name, count(*), avg(p.value), sum(p.value) from pmc_events p "
            "join kernels k on k.dispatch_id = p.dispatch_id group by 1, 2 order by 1, 2" % name_col).fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("\n# PMC counters (per-dispatch average, sum over dispatches)")
        for name, ctr, cnt, avg, tot in pmc:
            print("%-58s %-24s n=%-5d avg=%-16.1f sum=%.1f" % (name.split("(")[0][-58:], ctr, cnt, avg, tot))


if __name__ == "__main__":
    main(sys.argv[1])
