"""kernels + memory copies of the LAST h2agg_g1_msm call of a rocprofv3 --kernel-trace --memory-copy-trace run, on one time axis (us)"""
import csv, glob, sys
path = sys.argv[1]
ev = []
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("h2agg::", "").replace("void ", "")))
for f in glob.glob(path + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
# the last call: starts at the last large H2D copy that follows a gap of > 5 ms
starts = [i for i, e in enumerate(ev) if e[2].startswith("COPY") and "HOST_TO_DEVICE" in e[2] and (i == 0 or e[0] - max(x[1] for x in ev[max(0, i - 8):i]) > 5_000_000)]
i0 = starts[-1]
t0 = ev[i0][0]
for s, e, name in ev[i0:]:
    print("%9.1f %9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name[:70]))
