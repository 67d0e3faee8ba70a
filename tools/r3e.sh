mkdir -p gpurun_out/r3e; O=gpurun_out/r3e
python tools/pipeline_time.py 4 16 64 > $O/pipeline_time.txt 2>&1
H2AGG_TRACE_PHASES=1 python tools/pipeline_time.py 4 16 2>&1 | grep -A0 "phases" | awk 'NR%6==0' | head -40 > $O/phases.txt
H2AGG_HOST_SPONGE=scalar python tools/pipeline_time.py 4 2>&1 | grep host > $O/pipeline_scalar.txt
cat $O/pipeline_time.txt; cat $O/phases.txt; cat $O/pipeline_scalar.txt
