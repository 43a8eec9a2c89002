#!/bin/bash
mkdir -p gpurun_out/r3ts
for n in 4096 8192 16384 32768 65536; do for rr in 2 1; do
st=40; [ $n -ge 32768 ] && st=8
TD_ROW_ROUNDS=$rr timeout 600 python bench.py --n $n --steps $st --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r3ts/b.json 2> gpurun_out/r3ts/b.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3ts/b.json").read().strip().splitlines()[-1])
print("n=$n TD_ROW_ROUNDS=$rr", round(d["ms_per_step"],4), {k:(round(v["total_ms"],4),v["launches"]) for k,v in d["kernels"].items() if k in ("bid","sap")}, d["total_cost"])
PY
done; done
TD_ROW_ROUNDS=1 timeout 600 python tools/r3_bid0_check.py 2>&1 | tail -2
for w in g3 tick; do for rr in 2 1; do
TD_ROW_ROUNDS=$rr timeout 300 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', $rr, round(d['ms_per_step'],4))"
done; done
