#!/bin/bash
# Round-end evidence: rocprofv3 kernel-trace summary of the default bench command + PMC passes (each in its own run,
# never combined with other traces) for the dominant kernel.  Run on the GPU box from the repo root:
#   tools/profile_round.sh r01_final      -> gpurun_out/<tag>_*.txt  (copy the ones to keep into profiles/)
set -u
tag=${1:-r06_final}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cmd="python $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie-leg --agg-proofs 0"
cd /tmp
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o trace -- $cmd > $out/${tag}_bench_under_rocprof.json 2> /dev/null
python $root/tools/rocpd_summary.py /tmp/prof_kt/trace_results.db > $out/${tag}_kernel_stats.txt 2>&1
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY"; do
  name=(sq fetch write mem)
  rm -rf /tmp/prof_pmc && rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/prof_pmc -o pmc -- $cmd > /dev/null 2>&1
  python $root/tools/rocpd_summary.py /tmp/prof_pmc/pmc_results.db > $out/${tag}_pmc_${name[$i]}.txt 2>&1
  i=$((i+1))
done
cd $root
python tools/make_traffic_json.py $out/${tag} > $out/${tag}_traffic.json 2>> $out/${tag}_traffic.err
# the per-GPU share of BASELINE.json configs[4]: 16 x 2^22-point instance MSMs over one table (bench.py aggregate.config4_share)
cd /tmp
for pair in "FETCH_SIZE fetch" "WRITE_SIZE write"; do
  set -- $pair
  rm -rf /tmp/prof_pmcb && rocprofv3 --kernel-trace --pmc $1 -d /tmp/prof_pmcb -o pmc -- python $root/tools/fixed_base_big.py 22 --fixed-only > /dev/null 2>&1
  python $root/tools/rocpd_summary.py /tmp/prof_pmcb/pmc_results.db > $out/${tag}_pmc_batch_$2.txt 2>&1
done
cd $root
python tools/make_traffic_json.py $out/${tag} batch > $out/${tag}_batch_traffic.json 2>> $out/${tag}_traffic.err
# the same batch's kernels (the levels' sort: k_fb_partition / k_fb_bucket_sort; accumulation; the 16-grid tail), and the ordinary path beside it
cd /tmp && rm -rf /tmp/prof_fb && rocprofv3 --kernel-trace --stats -d /tmp/prof_fb -o fb -- python $root/tools/fixed_base_big.py 22 --fixed-only > $out/${tag}_fb_batch.txt 2>/dev/null
python $root/tools/rocpd_summary.py /tmp/prof_fb/fb_results.db > $out/${tag}_fb_kernel_stats.txt 2>&1
python $root/tools/r05_batch.py 22 16 > $out/${tag}_fb_vs_ordinary.txt 2>/dev/null
python $root/tools/r06_c_sweep.py > $out/${tag}_c_sweep.txt 2>/dev/null
cd $root
# batch kernels at n = 2^22 (HBM roofline rows)
cd /tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $root/tools/batch_roofline.py run > /dev/null 2>&1
python $root/tools/rocpd_summary.py /tmp/prof_b/b_results.db > $out/${tag}_batch_kernel_stats.txt 2>&1
python $root/tools/batch_roofline.py report $out/${tag}_batch_kernel_stats.txt > $out/${tag}_batch_roofline.txt 2>&1
cd $root
# full-size aggregation path trace
cd /tmp && rm -rf /tmp/prof_agg && rocprofv3 --kernel-trace --stats -d /tmp/prof_agg -o agg -- python $root/tools/agg_phases.py --reps 10 > $out/${tag}_agg_phases.txt 2>/dev/null
python $root/tools/rocpd_summary.py /tmp/prof_agg/agg_results.db > $out/${tag}_agg_kernel_stats.txt 2>&1
cd $root
# the from-bytes pipeline (h2agg_verify_aggregation) on both sponge backends + its phase split
python tools/pipeline_time.py 4 16 64 > $out/${tag}_pipeline_time.txt 2>&1
H2AGG_TRACE_PHASES=1 H2AGG_TRANSCRIPT=host python tools/pipeline_time.py 4 16 64 2>&1 | grep "phases" | awk 'NR%9==3' > $out/${tag}_pipeline_phases.txt
# kernel timeline of one evaluation (limb-parallel Horner chain)
bash tools/eval_trace.sh > /dev/null 2>&1; cp $out/eval_trace/eval_timeline.txt $out/${tag}_eval_timeline.txt 2>/dev/null
# kernel + copy timeline of one whole h2agg_verify_aggregation call (4 proofs) with its host phases
bash tools/pipeline_trace.sh > /dev/null 2>&1; (cat $out/pipeline_trace/timeline.txt; grep "h2agg phases" $out/pipeline_trace/phases.txt | tail -2 | cut -c1-900) > $out/${tag}_pipeline_trace.txt 2>/dev/null
# host-side phases of bench.py's aggregate leg (instance-column MSMs under the schema build, then the evaluation)
(python tools/agg_leg_phases.py --proofs 4; python tools/agg_leg_phases.py --proofs 16) > $out/${tag}_agg_leg_phases.txt 2>/dev/null
# the default bench line itself (no profiler attached); it reads the PMC evidence just collected from profiles/
mkdir -p profiles && cp $out/${tag}_traffic.json $out/${tag}_batch_traffic.json profiles/
python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
# everything to keep goes to profiles/ in ONE step (commit once):
mkdir -p profiles && cp $out/${tag}_* profiles/
ls -la $out | grep $tag
