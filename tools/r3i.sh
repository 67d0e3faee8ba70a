mkdir -p gpurun_out/r3i; O=gpurun_out/r3i
python -m pytest tests/test_gpu_verifier.py tests/test_ref_golden.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python tools/pipeline_time.py 4 16 64 > $O/pipeline_time.txt 2>&1
H2AGG_TRACE_PHASES=1 H2AGG_TRANSCRIPT=host python tools/pipeline_time.py 4 16 64 2>&1 | grep "phases" | awk 'NR%9==3' | head -12 > $O/phases.txt
grep -v device $O/pipeline_time.txt; cat $O/phases.txt
