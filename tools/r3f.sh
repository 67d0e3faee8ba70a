mkdir -p gpurun_out/r3f; O=gpurun_out/r3f
python tools/pipeline_time.py 4 16 64 > $O/pipeline_time.txt 2>&1
H2AGG_TRACE_PHASES=1 H2AGG_TRANSCRIPT=host python tools/pipeline_time.py 4 16 2>&1 | grep "phases" | awk 'NR%9==0' | head -12 > $O/phases.txt
grep -v device $O/pipeline_time.txt; cat $O/phases.txt
