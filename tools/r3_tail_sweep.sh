#!/bin/bash
mkdir -p gpurun_out/r3ts
run() {
env "$@" timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r3ts/b.json 2> gpurun_out/r3ts/b.err
python - "$*" <<PY
import json, sys
d=json.loads(open("gpurun_out/r3ts/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"],4), {k:(round(v["total_ms"],4),v["launches"]) for k,v in d["kernels"].items() if k in ("compress","bid","assign","sap")}, d["total_cost"])
PY
}
run TD_NOP=1
run TD_LDS_ROUNDS=2
run TD_LDS_ROUNDS=2 TD_LDS_GRID=2
run TD_ROW_ROUNDS=1
run TD_ROW_ROUNDS=3
run TD_MAX_ROUNDS=10
run TD_MAX_ROUNDS=8
run TD_MAX_ROUNDS=14
run TD_ROW_LOOP=3
run TD_ROW_LOOP=7
run TD_NOP=1
