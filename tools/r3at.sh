#!/bin/bash
run() { echo -n "$*: "; env "$@" H2AGG_TRACE_STREAMS=1 timeout 300 python tools/steps_time.py 20 40 2>&1 | grep "hold the main\|ms/step" | tail -3 | tr "\n" ";"; echo; }
run X=1
run H2AGG_NO_PLACE=1
run TORCH_STREAM=1
run TORCH_STREAM=1 H2AGG_NO_PLACE=1
run HIP_STREAM=nonblocking
run HIP_STREAM=blocking
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=8 TORCH_STREAM=1
timeout 300 python tools/pcie_rate.py 20 2>&1 | grep "page-locked"
timeout 300 python tools/pcie_rate.py 20 --torch-stream 2>&1 | grep "page-locked"
