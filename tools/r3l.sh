mkdir -p gpurun_out/r3l; O=gpurun_out/r3l
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pcie-leg --agg-proofs 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages_ms_per_step']
print('  ms/step %.3f  %.1f Mpts/s  acc_live %.3f |' % (d['ms_per_step'], d['value']/1e6, d['roofline']['avg_kernel_ms']), ' '.join('%s=%.3f' % (k.replace('msm_',''),v) for k,v in st.items()))"; }
for rep in 1 2; do
echo "default"; run
echo "glv 1"; run --glv 1
echo "glv 1 lpb 1"; run --glv 1 --lpb 1
echo "glv 1 lpb 2"; run --glv 1 --lpb 2
done > $O/glv.txt 2>&1
cat $O/glv.txt
