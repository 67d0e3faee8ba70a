#!/bin/bash
# kernel timeline of two consecutive MSMs inside a batch over a 2^22-point table: ordinary path, then fixed-base levels; run on the GPU box from the repo root
root=$(pwd); out=$root/gpurun_out/r05_trace; mkdir -p $out; rm -rf $out/*
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/pb && rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -- python $root/tools/r05_batch.py ${1:-22} 6 > $out/batch.txt 2>/dev/null
cd $root && python tools/timeline.py /tmp/pb 14 > $out/timeline_ordinary.txt 2>&1
python tools/timeline.py /tmp/pb 62 > $out/timeline_fixed.txt 2>&1
cat $out/batch.txt | tail -3; echo ORDINARY; cat $out/timeline_ordinary.txt; echo FIXED; cat $out/timeline_fixed.txt
