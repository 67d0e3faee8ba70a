mkdir -p gpurun_out/r3d; O=gpurun_out/r3d
python tools/pipeline_time.py 4 16 64 > $O/pipeline_time.txt 2>&1
python -m pytest tests/test_gpu_verifier.py tests/test_gpu_poseidon.py -x -q > $O/pytest_verifier.txt 2>&1
tail -30 $O/pipeline_time.txt; tail -5 $O/pytest_verifier.txt
