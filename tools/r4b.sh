#!/bin/bash
# round 4: kernel-level A/B of the lean accumulation variants (LDS-DMA prefetch) at 2 / 3 / 4 waves per SIMD
root=$(pwd); out=$root/gpurun_out/r4b; mkdir -p $out; rm -f $out/*
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_chain.py tests/test_gpu_overlap.py -m gpu -x -q 2>&1 | tail -3 > $out/pytest_dual.txt
H2AGG_ACC=lean1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -3 > $out/pytest_single.txt
cd /tmp
for m in lean2 lean1 generic lean2_3 lean1_3 lean2_2 lean1_2; do
  unset H2AGG_ACC H2AGG_ACC_LDS
  case $m in generic) export H2AGG_ACC=generic;; lean1*) export H2AGG_ACC=lean1;; esac
  case $m in *_3) export H2AGG_ACC_LDS=5000;; *_2) export H2AGG_ACC_LDS=12000;; esac
  python $root/tools/steps_time.py 20 40 > $out/${m}_steps_plain.txt 2>/dev/null
  rm -rf /tmp/p1 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o t -- python $root/tools/steps_time.py 20 30 > $out/${m}_steps.txt 2>/dev/null
  python $root/tools/rocpd_summary.py /tmp/p1/t_results.db 2>&1 | head -5 > $out/${m}_stats.txt
done
cd $root
cat $out/pytest_dual.txt $out/pytest_single.txt
for m in lean2 lean1 generic lean2_3 lean1_3 lean2_2 lean1_2; do echo "=== $m"; grep ms/step $out/${m}_steps_plain.txt | tail -2; cat $out/${m}_stats.txt | cut -c1-130 | sed -n '4,4p'; done
