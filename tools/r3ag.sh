#!/bin/bash
out=gpurun_out/r3ag; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests/test_gpu_verifier.py tests/test_gpu_poseidon.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
sed -i 's/for nthreads in (1, 2, 3, 4):/for nthreads in (1, 2, 4, 6, 8):/' tools/pipeline_concurrent.py
for k in 4 16; do timeout 400 python tools/pipeline_concurrent.py $k 3 2>&1 | grep -v amdgpu.ids >> $out/conc.txt; done
cat $out/conc.txt
