#!/bin/bash
# profiles/r06_gpu_suite.txt from the evidence leases (tools/evidence_job.sh) — run in the repo root after merging gpurun_out/
o=profiles/r06_gpu_suite.txt; d=gpurun_out/r06_final
{
echo "# Round 6: the GPU suite and the soaks of the FINAL tree, each block one fresh gpurun lease (one MI355X box, nothing kept between"
echo "# leases).  Complete output of \`python -m pytest tests/ -x -q -m gpu -rs\` of every lease, then the tallies of the stand-in soaks"
echo "# (tools/soak_standins.sh: fresh processes of tests/rccl_stub_ranks.py; tools/soak_bench_ranks.py: bench.py --gpus 2 over the"
echo "# library's communicator) and the offline fuzzers.  Collected: 375 GPU tests; 373 run on a one-GPU box (2 skipped: needs two"
echo "# GPUs; reference-made fixtures absent).  The sha256 in every block's header is over csrc/* + tests/cpp/rccl_stub.cpp."
echo "# (Logs of earlier leases of this round — round 5's stand-in under round 6's library, 150 of 150; mid-round soaks — went with the"
echo "# container they were kept in; their tallies are quoted in profiles/r06_sweeps.txt section 1 and not repeated here.)"
echo
for f in $d/suite_run*.txt; do cat $f; echo; done
for f in $d/soak_standins_lease*.txt; do echo "===== stand-in soak ($(basename $f .txt): behind that lease's suite run)"; cat $f; echo; done
[ -f $d/soak_bench_ranks.txt ] && { echo "===== bench.py --gpus 2 soak (behind a suite run, same lease)"; cat $d/soak_bench_ranks.txt; echo; }
[ -f $d/suite_durations.txt ] && { echo "===== a fifth fresh lease, the same tree: python -m pytest tests/ -x -q -m gpu --durations=60 (where the ten and a half minutes go)"; cat $d/suite_durations.txt; echo; }
for f in $d/fuzz_*.txt; do [ -f $f ] && { echo "===== $(basename $f .txt) (offline randomised differential run against the C oracle; command in its first line)"; cat $f; echo; }; done
} > $o
wc -l $o
