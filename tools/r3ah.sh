#!/bin/bash
out=gpurun_out/r3ah; mkdir -p $out; rm -f $out/*
timeout 900 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
tail -3 $out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3ah/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline'].get('traffic'), str(d['roofline'].get('pmc_evidence'))[:80])
print(d.get('pcie_inclusive'))
fp=d['aggregate']['full_pipeline']
print({k:fp[k] for k in ('proofs_per_sec','seconds_per_aggregation','recording_every_call','throughput_with_concurrent_contexts')})
print(fp['at_16_proofs_per_gpu'])
PY
