#!/bin/bash
out=$(pwd)/gpurun_out/r3ao; mkdir -p $out; root=$(pwd)
timeout 600 python -m pytest tests/test_gpu_schema.py tests/test_gpu_aggregate.py tests/test_gpu_verifier.py -x -q -m gpu 2>&1 | tail -2
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_ev
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ev -o ev -- python $root/tools/pipeline_time.py 4 > $out/pl.txt 2>&1
cd $root; python tools/eval_timeline.py /tmp/prof_ev > $out/eval_timeline.txt 2>&1
tail -8 $out/eval_timeline.txt
timeout 300 python tools/pipeline_time.py 4 16 2>&1 | grep "auto  "
