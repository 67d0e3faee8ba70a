#!/bin/bash
# profiles/r06_gpu_suite.txt from the three evidence leases (tools/evidence_job.sh) — run in the repo root after merging gpurun_out/
o=profiles/r06_gpu_suite.txt; d=gpurun_out/r06_final
{
echo "# Round 6: the GPU suite and the soaks of the FINAL tree, each block one fresh gpurun lease (one MI355X box, nothing kept between"
echo "# leases).  Complete output of \`python -m pytest tests/ -x -q -m gpu -rs\` three times, then the tallies of the stand-in soaks"
echo "# (tools/soak_standins.sh: fresh processes of tests/rccl_stub_ranks.py; tools/soak_bench_ranks.py: bench.py --gpus 2 over the"
echo "# library's communicator).  Collected: 375 GPU tests; 373 run on a one-GPU box (2 skipped: needs two GPUs; reference fixtures absent)."
echo
for r in 1 2 3; do cat $d/suite_run$r.txt; echo; done
echo "===== stand-in soak, lease 1 (behind suite run 1)"; cat $d/soak_standins.txt
echo; echo "===== stand-in soak, lease 3 (behind suite run 3)"; cat $d/soak_standins_lease3.txt
echo; echo "===== bench.py --gpus 2 soak, lease 2 (behind suite run 2)"; cat $d/soak_bench_ranks.txt
echo; echo "===== earlier in the round, tree of commit ec228ba + pool fix (before the download split), one lease: suite 372 passed, 2 skipped;"
cat gpurun_out/r06f/soak_standins_1.txt gpurun_out/r06f/soak_bench_ranks_1.txt
echo; echo "===== round 5's stand-in under round 6's library (reproduction attempt of GPUTEST_r05's failure), one lease"; cat gpurun_out/r06a/old_stub_tally.txt
echo; echo "===== tests/fuzz_fixed_base.py 300 61 (lease 3; small / carry / few-partition scalar patterns over c = 20 levels included)"; cat $d/fuzz_fixed_base.txt
} > $o
wc -l $o
