#!/bin/bash
out=gpurun_out/r3al; mkdir -p $out; rm -f $out/*
timeout 1200 python -m pytest tests/test_gpu_sort_dm.py tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_overlap.py tests/test_gpu_chain.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
for i in 1 2 3; do timeout 300 python tools/steps_time.py 20 40 2>&1 | grep ms/step | tail -1; done
timeout 300 python tools/steps_time.py 22 12 2>&1 | grep ms/step | tail -1
