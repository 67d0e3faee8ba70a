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
    # Kernels that run on a TAIL stream beside the next MSM's k_msm_accumulate (three VALU-saturating waves per SIMD): their
    # traced duration is mostly waiting for wave slots / issue cycles, not work.  They are marked '*' and left out of the
    # percentage column; `min_us` is what they take when they get the machine (DESIGN.md section 5 has the alone figures).
    waiting = ("k_msm_accumulate_big", "k_msm_accumulate_fix", "k_msm_big_combine", "k_msm_final", "k_msm_final_lp", "k_msm_reduce2d_parts", "k_msm_reduce2d_window",
               "k_msm_reduce_segments", "k_msm_window_sum", "k_msm_bucket_combine", "k_fb_fold", "k_fb_wsum")
    def waits(name):
        base = name.split("(")[0].replace("void ", "").replace("h2agg::", "").split("<")[0]
        return base in waiting
    total = sum(r[2] for r in rows if not waits(r[0])) or 1
    print("# kernel-trace summary of %s (durations in us)" % path)
    print("# '*': tail-stream kernels overlapped with the next MSM's accumulation: duration includes waiting beside it (not in pct)")
    print("%-58s %7s %12s %12s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, calls, tot, avg, mn, mx in rows:
        short = name.split("(")[0][-57:]
        w = waits(name)
        print("%-58s %7d %12.1f %12.2f %10.2f %10.2f %6s" % (("*" if w else "") + short, calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                             "-" if w else "%.2f" % (100.0 * tot / total)))
    try:
        # several rows per (dispatch, counter): one per hardware instance (XCD / SE) -> sum them per dispatch
        pmc = cur.execute(
            "select name, counter_name, count(*), avg(v), avg(d) from ("
            " select name, counter_name, dispatch_id, sum(counter_value) as v, max(duration) as d"
            " from pmc_events group by name, counter_name, dispatch_id) group by name, counter_name"
            " order by name, counter_name").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("\n# PMC counters: per-dispatch value (summed over hardware instances), averaged over dispatches")
        print("%-40s %-22s %6s %18s %12s" % ("kernel", "counter", "n", "avg_value", "avg_dur_us"))
        for name, ctr, cnt, avg, dur in pmc:
            if name.startswith("__amd") or "at::native" in name:
                continue
            print("%-40s %-22s %6d %18.1f %12.2f" % (name.split("(")[0][-40:], ctr, cnt, avg, (dur or 0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
