#!/bin/bash
out=gpurun_out/r3z; mkdir -p $out; rm -f $out/*
timeout 1200 python -m pytest tests/test_gpu_verifier.py tests/test_gpu_configs.py tests/test_gpu_aggregate.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
timeout 600 python tools/pipeline_time.py 4 16 64 2>&1 | grep -v amdgpu.ids > $out/pipeline.txt
cat $out/pipeline.txt
H2AGG_TRACE_PHASES=1 H2AGG_TRANSCRIPT=host timeout 300 python tools/pipeline_time.py 16 2>&1 | grep "phases" | sed -n '3p;12p' > $out/phases.txt
cat $out/phases.txt
