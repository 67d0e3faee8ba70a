"""timeline of the LAST evaluate_multiopen_proof of a rocprofv3 --kernel-trace CSV directory (tools/agg_phases.py under
rocprofv3 --kernel-trace --output-format csv -d DIR): python tools/eval_timeline.py DIR"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                     r["Kernel_Name"].split("(")[0].replace("h2agg::", "").replace("void ", ""), r.get("Queue_Id", "?")))
rows.sort()
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):   # (with --memory-copy-trace: the copies too)
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?"), "-"))
rows.sort()
end_marker = sys.argv[sys.argv.index("--end") + 1] if "--end" in sys.argv else "k_eval_tail_affine2"   # the launch the window ends at
i1 = max(i for i, r in enumerate(rows) if r[2].startswith(end_marker))
if "--last" in sys.argv:    # the N launches in front of the last evaluation's end (a whole h2agg_verify_aggregation call)
    i0 = max(0, i1 - int(sys.argv[sys.argv.index("--last") + 1]))
else:
    i0 = max(i for i, r in enumerate(rows[:i1]) if r[2].startswith("k_tape_load_consts") or r[2].startswith("k_tape_run_lds"))
t0 = rows[i0][0]
busy_end, idle = t0, 0
for s, e, n, q in rows[i0:i1 + 1]:
    if s > busy_end:
        idle += s - busy_end
    busy_end = max(busy_end, e)
    print("%8.1f %8.1f %7.1f q%-2s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:50]))
print("span %.1f us, device idle inside it %.1f us" % ((rows[i1][1] - t0) / 1e3, idle / 1e3))
