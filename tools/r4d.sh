#!/bin/bash
# round 4: alternating A/B on one box: generic (env), lean early-issue, lean late-issue
root=$(pwd); out=$root/gpurun_out/r4d; mkdir -p $out; rm -f $out/*
P=halo2-snark-aggregator_amd
cp $P/libh2agg.so /tmp/keep.so
for round in 1 2 3 4; do
  for v in generic early late; do
    unset H2AGG_ACC
    case $v in generic) cp tools/ab/lean_late.so $P/libh2agg.so; export H2AGG_ACC=generic;; early) cp tools/ab/lean_early.so $P/libh2agg.so;; late) cp tools/ab/lean_late.so $P/libh2agg.so;; esac
    echo "$v $(python tools/steps_time.py 20 40 2>/dev/null | grep ms/step | awk '{print $2}' | tr '\n' ' ')" >> $out/ab.txt
  done
done
cp /tmp/keep.so $P/libh2agg.so
cat $out/ab.txt
