#!/bin/bash
timeout 300 python tools/steps_time.py 20 40 2>&1 | grep ms/step | tail -1
timeout 300 python tools/pcie_rate.py 20 2>&1 | grep "page-locked"
timeout 300 python tools/pipeline_time.py 4 16 2>&1 | grep "auto  "
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); fp=d['aggregate']['full_pipeline']; print('bench:', d['value'], d['ms_per_step'], d['pcie_inclusive']['ms_per_msm'], d['aggregate']['proofs_per_sec'], fp['proofs_per_sec'], fp['at_16_proofs_per_gpu']['proofs_per_sec'], [x['proofs_per_sec'] for x in fp['throughput_with_concurrent_contexts']])"
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_verifier.py tests/test_gpu_overlap.py -x -q -m gpu 2>&1 | tail -2
