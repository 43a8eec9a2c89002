#!/bin/bash
mkdir -p gpurun_out/r3b0
run() {
env "$@" TD_BID0=1 timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3b0/b.json 2> gpurun_out/r3b0/b.err
python - "$*" <<PY
import json, sys
d=json.loads(open("gpurun_out/r3b0/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], {k:(round(v["total_ms"],4),v["launches"]) for k,v in d["kernels"].items() if k in ("compress","bid")}, d["total_cost"])
PY
}
run TD_BID0_SHAPE=0
run TD_BID0_SHAPE=0 TD_BID0_GRID=3
run TD_BID0_SHAPE=0 TD_BID0_GRID=4
run TD_BID0_SHAPE=0 TD_BID0_GRID=1
run TD_BID0_SHAPE=1 TD_BID0_GRID=1
run TD_BID0_SHAPE=1 TD_BID0_GRID=2
run TD_BID0_SHAPE=0 TD_BID0_GRID=8
run2() {
env "$@" TD_BID0=0 timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3b0/b.json 2> gpurun_out/r3b0/b.err
python - "$*" <<PY
import json, sys
d=json.loads(open("gpurun_out/r3b0/b.json").read().strip().splitlines()[-1])
print("base", sys.argv[1], d["ms_per_step"], {k:(round(v["total_ms"],4),v["launches"]) for k,v in d["kernels"].items() if k in ("compress","bid")}, d["total_cost"])
PY
}
run2 TD_CREG_SHAPE=0
run2 TD_CREG_SHAPE=1
run2 TD_CREG_SHAPE=1 TD_BID0_GRID=3
run2 TD_CREG_SHAPE=1 TD_BID0_GRID=4
run2 TD_CREG_SHAPE=2 TD_BID0_GRID=1
run2 TD_CREG_SHAPE=2 TD_BID0_GRID=2
