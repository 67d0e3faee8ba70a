mkdir -p gpurun_out/r3p; O=gpurun_out/r3p
for a in "--agg-proofs 0" ""; do
python bench.py --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench $a: value %.1f pcie %.2f ms' % (d['value']/1e6, d['pcie_inclusive']['ms_per_msm']))"
done
