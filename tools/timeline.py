"""print a per-kernel timeline (start / end / duration, us, relative) of a window of a rocprofv3 --kernel-trace CSV"""
import csv, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("h2agg::", "").replace("void ", ""), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
acc = [i for i, r in enumerate(rows) if r[2].startswith("k_msm_accumulate") and not r[2].startswith("k_msm_accumulate_big")]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(acc) - 6
i0 = acc[skip]
i1 = acc[skip + 2]
t0 = rows[i0][0]
for s, e, name, q, st in rows[i0:i1 + 12]:
    print("%9.1f %9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name[:60]))
