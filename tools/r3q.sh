mkdir -p gpurun_out/r3q; O=gpurun_out/r3q
python -m pytest tests/test_gpu_sort_dm.py -x -q -k c17 > $O/pytest_c17.txt 2>&1; tail -5 $O/pytest_c17.txt
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pcie-leg --agg-proofs 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages_ms_per_step']
print('  ms/step %.3f  %.1f Mpts/s  acc_live %.3f |' % (d['ms_per_step'], d['value']/1e6, d['roofline']['avg_kernel_ms']), ' '.join('%s=%.3f' % (k.replace('msm_',''),v) for k,v in st.items()))"; }
for rep in 1 2 3; do echo "c16"; run; echo "c17"; run --window 17; done > $O/ab_c17.txt 2>&1; cat $O/ab_c17.txt
