#!/bin/bash
# A/B of the bucket-accumulation kernels on ONE box (run on the GPU box from the repo root, after
# `python halo2-snark-aggregator_amd/build_ext.py --measure` on the build host): alternates the 166-VGPR kernel, the lean
# single-chain and the lean dual-chain kernel (H2AGG_ACC, a -DH2AGG_MEASURE_KNOBS switch) at full occupancy and with the
# occupancy capped by unused LDS (H2AGG_ACC_LDS), steps_time + rocprofv3 kernel trace.  Output: gpurun_out/ab_accumulate/.
root=$(pwd); out=$root/gpurun_out/ab_accumulate; mkdir -p $out; rm -f $out/*
P=halo2-snark-aggregator_amd
[ -f tools/libh2agg_measure.so ] || { echo "build tools/libh2agg_measure.so first (build_ext.py --measure)"; exit 1; }
cp $P/libh2agg.so /tmp/keep.so; cp tools/libh2agg_measure.so $P/libh2agg.so
export TMPDIR=/tmp
for round in 1 2 3; do
  for m in generic lean1 lean2 lean2_3waves lean2_2waves; do
    unset H2AGG_ACC H2AGG_ACC_LDS
    case $m in generic) export H2AGG_ACC=generic;; lean1) export H2AGG_ACC=lean1;; esac
    case $m in *_3waves) export H2AGG_ACC_LDS=13000;; *_2waves) export H2AGG_ACC_LDS=20000;; esac
    echo "$m $(python tools/steps_time.py 20 40 2>/dev/null | grep ms/step | awk '{print $2}' | tr '\n' ' ')" >> $out/steps.txt
    if [ $round = 1 ]; then
      (cd /tmp && rm -rf /tmp/p1 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t -- python $root/tools/steps_time.py 20 30 > /dev/null 2>&1
       python $root/tools/rocpd_summary.py /tmp/p1/t_results.db 2>&1 | sed -n '4,4p' | cut -c1-130 | sed "s/^/$m  /" >> $out/kernel.txt)
    fi
  done
done
cp /tmp/keep.so $P/libh2agg.so
cat $out/steps.txt $out/kernel.txt
