mkdir -p gpurun_out/r3u; O=$(pwd)/gpurun_out/r3u; root=$(pwd)
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_ev && H2AGG_TRANSCRIPT=host rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ev -o ev -- python $root/tools/pipeline_time.py 4 > $O/pl.txt 2>&1
cd $root
python tools/eval_timeline.py /tmp/prof_ev > $O/eval_timeline.txt 2>&1
cat $O/eval_timeline.txt
python - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("/tmp/prof_ev/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("h2agg::","").replace("void ","")))
rows.sort()
# the last full pipeline call: from the last k_transcript_elements back to the previous instance MSM start
i1=max(i for i,r in enumerate(rows) if r[2].startswith("k_eval_tail_affine2"))
i0=max(i for i,r in enumerate(rows[:i1]) if r[2].startswith("k_transcript_elements"))
# go back to first kernel after previous eval tail
ip=max(i for i,r in enumerate(rows[:i0]) if r[2].startswith("k_eval_tail_affine2"))
t0=rows[ip+1][0]
print("== one whole h2agg_verify_aggregation (4 proofs), kernels before the evaluation")
for s,e,n in rows[ip+1:i0+3]:
    print("%8.1f %8.1f %7.1f %s" % ((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,n[:60]))
PY
