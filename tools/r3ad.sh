#!/bin/bash
out=gpurun_out/r3ad; mkdir -p $out; rm -f $out/*
timeout 1500 python -m pytest tests/test_gpu_sort_dm.py tests/test_gpu_overlap.py tests/test_gpu_scale.py tests/test_gpu_chain.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
for l in 20 21 22; do echo "== 2^$l default" >> $out/steps.txt; timeout 300 python tools/steps_time.py $l 12 2>&1 | grep ms/step | tail -2 >> $out/steps.txt; done
cat $out/steps.txt
echo "== pcie 22, 24" >> $out/pcie.txt
for l in 22 24; do timeout 300 python tools/pcie_rate.py $l 2>&1 | grep "n=2" >> $out/pcie.txt; done
cat $out/pcie.txt
