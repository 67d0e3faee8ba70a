#!/bin/bash
# One lease of the round-end evidence: the complete `pytest tests/ -x -q -m gpu -rs` output of this tree, then
# the job named by $2 (standins | bench | profile | none).   usage: tools/evidence_job.sh <run number> <standins|bench|profile|none> [soak counts]
run=$1; what=${2:-none}
out=gpurun_out/r06_final; mkdir -p $out
sha=$(cat halo2-snark-aggregator_amd/csrc/*.h* halo2-snark-aggregator_amd/csrc/*.inc halo2-snark-aggregator_amd/csrc/*.hip tests/cpp/rccl_stub.cpp | sha256sum | cut -c1-16)
{ echo "===== fresh lease $run: python -m pytest tests/ -x -q -m gpu -rs    (csrc + stand-in sha256 $sha; $(date -u +%FT%TZ); $(hostname))"; python -m pytest tests/ -x -q -m gpu -rs 2>&1; echo "===== exit code $?"; } > $out/suite_run$run.txt 2>&1
tail -4 $out/suite_run$run.txt
case $what in
  standins) tools/soak_standins.sh ${3:-150} ${4:-40} $out/soak_standins_lease$run.txt ;;
  bench) python tools/soak_bench_ranks.py ${3:-50} $out/soak_bench_ranks.txt ;;
  profile) bash tools/profile_round.sh r06_final > $out/profile_round.log 2>&1; tail -3 $out/profile_round.log ;;
esac
