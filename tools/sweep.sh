#!/bin/bash
# parameter sweep helper: tools/sweep.sh "<bench args>" ...   (STEPS=30 by default)
for a in "$@"; do
  python bench.py --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline --no-pcie-leg --agg-proofs 0 $a 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages_ms_per_step']
print('$a', '| ms/step %.3f  %.1f Mpts/s  acc_live %.3f |' % (d['ms_per_step'], d['value']/1e6, d['roofline']['avg_kernel_ms']), ' '.join('%s=%.3f' % (k.replace('msm_',''),v) for k,v in st.items()))"
done
