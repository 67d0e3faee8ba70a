#!/bin/bash
for q in 4 8 4 8 16; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 300 python tools/steps_time.py 20 40 2>&1 | grep ms/step | tail -2; done
for q in 4 8; do echo "== bench GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); fp=d['aggregate']['full_pipeline']; print('bench:', d['value'], d['pcie_inclusive']['ms_per_msm'], d['aggregate']['proofs_per_sec'], fp['proofs_per_sec'], fp['at_16_proofs_per_gpu']['proofs_per_sec'], [x['proofs_per_sec'] for x in fp['throughput_with_concurrent_contexts']])"; done
