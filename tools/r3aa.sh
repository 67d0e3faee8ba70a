#!/bin/bash
out=gpurun_out/r3aa; mkdir -p $out; rm -f $out/*
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
tail -3 $out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3aa/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','roofline')})
print(d.get('pcie_inclusive'))
a=d.get('aggregate',{})
fp=a.get('full_pipeline')
print(json.dumps(fp,indent=1)[:1800] if fp else a)
PY
