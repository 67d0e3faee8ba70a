#!/bin/bash
for q in 4 8 16 24; do for ts in "" 1; do echo -n "GPU_MAX_HW_QUEUES=$q TORCH_STREAM=$ts: "; GPU_MAX_HW_QUEUES=$q TORCH_STREAM=$ts timeout 300 python tools/steps_time.py 20 40 2>&1 | grep ms/step | tail -1; done; done
