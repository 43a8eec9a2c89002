#!/bin/bash
mkdir -p gpurun_out/r3n6
run() {
env "$@" timeout 600 python bench.py --n 65536 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3n6/b.json 2> gpurun_out/r3n6/b.err
python - "$*" <<PY
import json, sys
d=json.loads(open("gpurun_out/r3n6/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"],3), {k:round(v["total_ms"],3) for k,v in d["kernels"].items() if k in ("gen","compress","bid","sap")}, d["total_cost"])
PY
}
run TD_GEN_GRID=128
run TD_GEN_GRID=256
run TD_GEN_GRID=512
run TD_GEN_GRID=1024
run TD_GEN_GRID=2048
run TD_GEN_GRID=64
