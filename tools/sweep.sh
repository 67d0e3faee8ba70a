#!/bin/bash
# parameter sweep helper: tools/sweep.sh "<bench args>" ...
for a in "$@"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --agg-proofs 0 $a 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages_ms_per_step']
print('$a', '| ms/step %.3f' % d['ms_per_step'], ' '.join('%s=%.3f' % (k.replace('msm_',''),v) for k,v in st.items()))"
done
