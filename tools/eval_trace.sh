#!/bin/bash
# kernel timeline of one evaluate_multiopen_proof (4 proofs) + the from-bytes pipeline timings; run on the GPU box from the repo root
root=$(pwd); out=$root/gpurun_out/eval_trace; mkdir -p $out; rm -rf $out/*
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/pe && rocprofv3 --kernel-trace --output-format csv -d /tmp/pe -- python $root/tools/agg_phases.py --reps 3 > $out/agg_phases.txt 2>/dev/null
cd $root && python tools/eval_timeline.py /tmp/pe > $out/eval_timeline.txt 2>&1
python tools/pipeline_time.py 4 16 > $out/pipeline_time.txt 2>&1
tail -45 $out/eval_timeline.txt; grep "auto " $out/pipeline_time.txt | head -8
