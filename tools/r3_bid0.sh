#!/bin/bash
mkdir -p gpurun_out/r3b0
timeout 900 python tools/r3_bid0_check.py > gpurun_out/r3b0/check.log 2>&1; echo "check rc $?" >> gpurun_out/r3b0/check.log
tail -12 gpurun_out/r3b0/check.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3b0/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3b0/tests.log
tail -4 gpurun_out/r3b0/tests.log
for m in 1 0; do
TD_BID0=$m timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3b0/bench_bid0_$m.json 2> gpurun_out/r3b0/bench_$m.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3b0/bench_bid0_$m.json").read().strip().splitlines()[-1])
print("TD_BID0=$m", d["ms_per_step"], d["value"], {k:(round(v["total_ms"],4),v["launches"]) for k,v in d["kernels"].items()}, d["roofline"]["frac"])
PY
done
