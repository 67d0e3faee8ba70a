#!/bin/bash
# accumulation pinned at 2 waves per SIMD (256-thread workgroups + unused LDS: 2 x 56000 B fit a CU, 3 do not) with the next sort beside it (overlap level 3)
out=gpurun_out/r3v; mkdir -p $out; rm -f $out/steps2.txt
run() { echo "== $*" >> $out/steps2.txt; env "$@" timeout 120 python tools/steps_time.py 20 40 2>&1 | grep ms/step >> $out/steps2.txt; }
run LEVEL=2
run LEVEL=2 H2AGG_ACC_BLOCK=256 H2AGG_ACC_LDS=56000
run LEVEL=3 H2AGG_ACC_BLOCK=256 H2AGG_ACC_LDS=56000
run LEVEL=3 WINDOW=17 H2AGG_ACC_BLOCK=256 H2AGG_ACC_LDS=56000
run LEVEL=2 H2AGG_ACC_BLOCK=256 H2AGG_ACC_LDS=41000
cat $out/steps2.txt
