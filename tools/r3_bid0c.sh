#!/bin/bash
mkdir -p gpurun_out/r3b0
for m in 1 0; do
TD_BID0=$m timeout 600 python bench.py --n 65536 --steps 8 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r3b0/b65_$m.json 2> gpurun_out/r3b0/b65_$m.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3b0/b65_$m.json").read().strip().splitlines()[-1])
print("n=65536 TD_BID0=$m", d["ms_per_step"], {k:(round(v["total_ms"],4),v["launches"]) for k,v in d["kernels"].items() if k in ("compress","bid","sap")}, d["total_cost"])
PY
done
for n in 9000 12288 20000 32768; do for m in 1 0; do
TD_BID0=$m timeout 600 python bench.py --n $n --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r3b0/bn.json 2> gpurun_out/r3b0/bn.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3b0/bn.json").read().strip().splitlines()[-1])
print("n=$n TD_BID0=$m", d["ms_per_step"], {k:(round(v["total_ms"],4),v["launches"]) for k,v in d["kernels"].items() if k in ("compress","bid")}, d["total_cost"])
PY
done; done
