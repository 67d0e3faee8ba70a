#!/bin/bash
timeout 300 python tools/pcie_rate.py 20 2>&1 | grep "page-locked"
for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench:', d['value'], d['pcie_inclusive']['ms_per_msm'], d['aggregate']['full_pipeline']['proofs_per_sec'])"; done
