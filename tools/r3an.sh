#!/bin/bash
P=halo2-snark-aggregator_amd
cp $P/libh2agg.so /tmp/base.so
for rep in 1 2 3; do
  cp /tmp/base.so $P/libh2agg.so; echo -n "3 waves (166 VGPRs, 68 B scratch): "; timeout 300 python tools/steps_time.py 20 40 2>&1 | grep ms/step | tail -1
  cp tools/_ab/w4.so $P/libh2agg.so; echo -n "4 waves (128 VGPRs, 232 B scratch): "; timeout 300 python tools/steps_time.py 20 40 2>&1 | grep ms/step | tail -1
done
cp tools/_ab/w4.so $P/libh2agg.so; timeout 300 python -m pytest tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -1
cp /tmp/base.so $P/libh2agg.so
