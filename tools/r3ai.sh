#!/bin/bash
out=gpurun_out/r3ai; mkdir -p $out; rm -f $out/*
run() { echo "== $*" >> $out/pcie.txt; env "${@:2}" timeout 300 python tools/pcie_rate.py 20 $1 2>&1 | grep "page-locked" >> $out/pcie.txt; }
for x in 0 1 2 3 4 5; do run --extra-streams=$x X=1; done
run --extra-streams=1 GPU_MAX_HW_QUEUES=8
run --extra-streams=2 GPU_MAX_HW_QUEUES=8
run --extra-streams=3 GPU_MAX_HW_QUEUES=16
cat $out/pcie.txt
GPU_MAX_HW_QUEUES=8 timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench with GPU_MAX_HW_QUEUES=8:', d['value'], d['pcie_inclusive']['ms_per_msm'], d['aggregate']['full_pipeline']['proofs_per_sec'])"
