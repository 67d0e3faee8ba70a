#!/bin/bash
# chaos runs: delays in front of tails (1), accumulations (2), sorts (4)
for bits in 1 2 4 3 5; do
  echo "== H2AGG_CHAOS=$bits"
  H2AGG_CHAOS=$bits timeout 900 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_aggregate.py tests/test_gpu_configs.py tests/test_gpu_schema.py tests/test_gpu_chain.py -x -q -m gpu 2>&1 | tail -6
done
