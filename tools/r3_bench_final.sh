#!/bin/bash
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo rc $?
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
print({k:(round(v["ms_per_step"],3) if "ms_per_step" in v else "-") for k,v in d["other_workloads"].items()})
PY
tail -3 gpurun_out/bench_final.err
