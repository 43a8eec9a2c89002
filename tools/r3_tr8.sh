#!/bin/bash
mkdir -p gpurun_out/r3tr8
for r in 1024 512 256 1024 512; do
TD_TR8_ROWS=$r timeout 300 python bench.py --workload g3 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3tr8/b.json 2> gpurun_out/r3tr8/b.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3tr8/b.json").read().strip().splitlines()[-1])
print("TD_TR8_ROWS=$r", round(d["ms_per_step"],4), {k:(round(v["total_ms"],4),v["launches"]) for k,v in d["kernels"].items() if k in ("compress","bid","cost_build")}, d["total_cost"])
PY
done
