#!/bin/bash
# kernel timeline of the LAST h2agg_verify_aggregation call of tools/pipeline_one.py (4 proofs, default switches) + its host phases; GPU box, repo root
root=$(pwd); out=$root/gpurun_out/pipeline_trace; mkdir -p $out; rm -rf $out/*
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/pt && H2AGG_TRACE_PHASES=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pt -- python $root/tools/pipeline_one.py 4 > $out/ab.txt 2> $out/phases.txt
cd $root && python tools/eval_timeline.py /tmp/pt --last 46 > $out/timeline.txt 2>&1
tail -75 $out/timeline.txt; grep "h2agg phases" $out/phases.txt | tail -3
