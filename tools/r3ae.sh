#!/bin/bash
out=gpurun_out/r3ae; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_gpu_verifier.py tests/test_gpu_configs.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -6 $out/pytest.txt
timeout 600 python tools/pipeline_time.py 4 16 64 2>&1 | grep -v amdgpu.ids | grep "auto \|host threads\|recorded" > $out/pipeline.txt
cat $out/pipeline.txt
