#!/bin/bash
out=$(pwd)/gpurun_out/r3x; mkdir -p $out; root=$(pwd)
export TMPDIR=/tmp
cd /tmp
for cfg in "5 1"; do
  set -- $cfg
  rm -rf /tmp/prof_pc
  H2AGG_PCIE_SLICES=$1 H2AGG_PCIE_GLV=$2 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_pc -o pc -- python $root/tools/pcie_one.py 20 > $out/one_$1_$2.txt 2>&1
  python $root/tools/pcie_timeline.py /tmp/prof_pc > $out/timeline_$1_$2.txt 2>&1
done
ls /tmp/prof_pc | head
cat $out/timeline_5_1.txt | head -150
