#!/bin/bash
out=gpurun_out/r3af; mkdir -p $out; rm -f $out/*
for n in 4 16; do
H2AGG_TRACE_PHASES=1 H2AGG_TRANSCRIPT=host timeout 300 python tools/pipeline_time.py $n 2>&1 | grep "phases" | sed -n '25,27p' >> $out/phases.txt
H2AGG_COMB_MSM=0 H2AGG_TRACE_PHASES=1 H2AGG_TRANSCRIPT=host timeout 300 python tools/pipeline_time.py $n 2>&1 | grep "phases" | sed -n '25,27p' >> $out/phases.txt
done
cat $out/phases.txt
