#!/bin/bash
out=gpurun_out/r3ab; mkdir -p $out; rm -f $out/*
run() { echo "== $*" >> $out/pcie.txt; env "${@:2}" timeout 300 python tools/pcie_rate.py 20 $1 2>&1 | grep -v amdgpu.ids | grep "page-locked" >> $out/pcie.txt; }
run "" H2AGG_PCIE_CHAIN=1
run --torch-stream H2AGG_PCIE_CHAIN=1
run --torch-stream H2AGG_PCIE_CHAIN=0
cat $out/pcie.txt
for i in 1 2; do timeout 300 python tools/steps_time.py 20 40 2>&1 | grep ms/step | tail -1; done
timeout 600 python bench.py --no-cpu-baseline --agg-proofs 0 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d.get('pcie_inclusive',{}).get('ms_per_msm'))"
