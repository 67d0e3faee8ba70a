#!/bin/bash
# chained slices (one bucket set, one tail) for the host-buffer MSM: parity, then the PCIe-inclusive time per slice count
out=gpurun_out/r3w; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_overlap.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
run() { echo "== $*" >> $out/pcie.txt; env "$@" timeout 120 python tools/pcie_rate.py 20 2>&1 | grep -v amdgpu.ids | grep page-locked >> $out/pcie.txt; }
run H2AGG_PCIE_CHAIN=0
for j in 3 4 5 6 8; do
  run H2AGG_PCIE_SLICES=$j H2AGG_PCIE_GLV=1
done
run H2AGG_PCIE_SLICES=4 H2AGG_PCIE_GLV=-1
run H2AGG_PCIE_SLICES=6 H2AGG_PCIE_GLV=-1
cat $out/pcie.txt
