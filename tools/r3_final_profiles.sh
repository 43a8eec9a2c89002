#!/bin/bash
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_r3z_g3 gpurun_out/prof_r3z_tick
STEPS=20 PMC=1 bash tools/profile_round.sh r3z_g3 g3 > /dev/null 2>&1
STEPS=50 PMC=0 bash tools/profile_round.sh r3z_tick tick > /dev/null 2>&1
find gpurun_out/prof_r3z_g3 gpurun_out/prof_r3z_tick -name "*.csv" -size +8M -delete
head -16 gpurun_out/prof_r3z_g3/summary.txt; grep -A8 "FETCH_KB" gpurun_out/prof_r3z_g3/summary.txt | head -10; tail -2 gpurun_out/prof_r3z_g3/summary.txt | cut -c1-400
head -22 gpurun_out/prof_r3z_tick/summary.txt; tail -2 gpurun_out/prof_r3z_tick/summary.txt | cut -c1-300
mkdir -p gpurun_out/r3fin
timeout 900 python bench.py > gpurun_out/r3fin/bench_default.json 2> gpurun_out/r3fin/bench_default.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3fin/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"], d["cpu_baseline"])
print({k:(v.get("ms_per_step") or v.get("ms") or v) for k,v in d.get("other_workloads",{}).items() if isinstance(v,dict)})
PY
