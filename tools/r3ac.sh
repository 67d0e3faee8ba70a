#!/bin/bash
out=gpurun_out/r3ac; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
for w in 16 17; do echo "== 2^22 window $w" >> $out/steps22.txt; WINDOW=$w timeout 300 python tools/steps_time.py 22 12 2>&1 | grep ms/step >> $out/steps22.txt; done
for w in 16 17; do echo "== 2^21 window $w" >> $out/steps22.txt; WINDOW=$w timeout 300 python tools/steps_time.py 21 20 2>&1 | grep ms/step >> $out/steps22.txt; done
cat $out/steps22.txt
