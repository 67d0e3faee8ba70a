#!/bin/bash
out=gpurun_out/r3y; mkdir -p $out; rm -f $out/*
run() { echo "== $*" >> $out/pcie.txt; env "${@:2}" timeout 300 python tools/pcie_rate.py $1 2>&1 | grep -v amdgpu.ids | grep "n=2" >> $out/pcie.txt; }
for l in 17 18 19 20 21 22 24; do
  run $l H2AGG_PCIE_CHAIN=0
  run $l H2AGG_PCIE_CHAIN=1
done
cat $out/pcie.txt
