#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r4e; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests/test_gpu_comm_group.py tests/test_gpu_sharded.py tests/test_gpu_comm.py -m gpu -x -q 2>&1 | tail -8 > $out/pytest.txt
timeout 900 python bench.py --no-cpu-baseline --no-pcie-leg > $out/bench.json 2> $out/bench.err
cat $out/pytest.txt; tail -3 $out/bench.err; python - <<PY
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value']/1e6, d['value_no_spinup']/1e6)
a=d.get('aggregate',{})
print({k:(v if not isinstance(v,dict) else '...') for k,v in a.items()})
for k in ('config3','config4_share'):
    print(k, json.dumps(a.get(k), indent=1)[:1500])
PY
