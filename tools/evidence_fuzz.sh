#!/bin/bash
# One lease behind the three suite leases: the offline fuzzers against the C oracle and a second default bench line (another box).
out=gpurun_out/r06_final; mkdir -p $out
run() { f=$out/$1.txt; shift; { echo "\$ $*   ($(date -u +%FT%TZ))"; timeout 900 "$@" 2>&1 | grep -v amdgpu.ids | tail -6; echo "exit code ${PIPESTATUS[0]}"; } > $f; tail -2 $f; }
run fuzz_fixed_base python tests/fuzz_fixed_base.py 270 61
run fuzz_msm python tests/fuzz_msm.py --seconds 150 --seed 6
run fuzz_schema python tests/fuzz_schema.py
python bench.py > $out/bench_default_second_box.json 2> $out/bench_default_second_box.err; head -c 300 $out/bench_default_second_box.json
