#!/bin/bash
bash tools/profile_round.sh r03_final > /dev/null 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/r03_final_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["traffic"], d["roofline"]["avg_kernel_ms"], d["roofline"]["achieved"])
v=d["roofline"]["valu"]; print(v["cycles_per_instruction_per_simd"], v["shader_clock_ghz"], v["pmc_run_kernel_ms"], v["issue_frac"], v["issue_frac_at_measured_rates"])
print(d["pcie_inclusive"]["ms_per_msm"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("fair_cpu_pippenger",{}).get("value"), d["cpu_baseline"].get("fair_cpu_pippenger",{}).get("seconds"))
a=d["aggregate"]; fp=a["full_pipeline"]
print(a["proofs_per_sec"], a["seconds_per_aggregation"], a["at_16_proofs_per_gpu"]["proofs_per_sec"], a["at_16_proofs_per_gpu"]["seconds_per_aggregation"], a.get("cpu_baseline_aggregate",{}).get("value"))
print(fp["proofs_per_sec"], fp["seconds_per_aggregation"], fp["at_16_proofs_per_gpu"]["proofs_per_sec"], fp["at_16_proofs_per_gpu"]["seconds_per_aggregation"], [x["proofs_per_sec"] for x in fp["throughput_with_concurrent_contexts"]])
PY
sed -n 4p gpurun_out/r03_final_kernel_stats.txt; grep -v amdgpu gpurun_out/r03_final_pipeline_time.txt | grep "auto  .*with"
