#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_lib.sh <other.so> [bench args]
# runs bench.py with the in-tree libh2agg.so, then with <other.so> in its place (plus the MSM parity tests), twice each
P=halo2-snark-aggregator_amd
other=$1; shift
run() {
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pcie-leg --agg-proofs 0 "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['roofline']['stages_ms_per_step']
print('  ms/step %.3f  %.1f Mpts/s  acc_live %.3f |' % (d['ms_per_step'], d['value']/1e6, d['roofline']['avg_kernel_ms']), ' '.join('%s=%.3f' % (k.replace('msm_',''),v) for k,v in st.items()))"
}
cp $P/libh2agg.so /tmp/base.so
for rep in 1 2; do
  echo "base"; cp /tmp/base.so $P/libh2agg.so; run "$@"
  echo "other ($other)"; cp $other $P/libh2agg.so; run "$@"
done
echo "parity with $other:"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_scale.py -q -x 2>&1 | tail -3
cp /tmp/base.so $P/libh2agg.so
