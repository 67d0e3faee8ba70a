run() { # env, args
  out=$(env $1 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pcie-leg --agg-proofs 0 $2 2>&1 | tail -1)
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages_ms_per_step']
print('$1 $2', '| ms/step %.3f  %.1f Mpts/s  acc_live %.3f |' % (d['ms_per_step'], d['value']/1e6, d['roofline']['avg_kernel_ms']), ' '.join('%s=%.3f' % (k.replace('msm_',''),v) for k,v in st.items()))"
}
run X=1 ""
run H2AGG_PAR4=-1 ""
run H2AGG_PAR4=-1 "--seg 64"
run H2AGG_PAR4=-1 "--seg 16"
run H2AGG_PAR4=1 "--seg 64"
run H2AGG_PAR4=1 "--seg 16"
run X=1 ""
