#!/bin/bash
mkdir -p gpurun_out/r3f2
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3f2/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3f2/tests.log
tail -3 gpurun_out/r3f2/tests.log
timeout 400 python tools/gpu_stress_large.py 23 300 > gpurun_out/r3f2/stress_large.log 2>&1
tail -2 gpurun_out/r3f2/stress_large.log | cut -c1-200
grep -v "^ok" gpurun_out/r3f2/stress_large.log | head -5
python - <<PY
import re
ts=[]
for l in open("gpurun_out/r3f2/stress_large.log"):
    m=re.match(r"ok\s+(\S+)\s+(\d+)\s+([\d.]+) ms", l)
    if m: ts.append((float(m.group(3)), m.group(1), int(m.group(2))))
ts.sort(reverse=True)
print("slowest:", ts[:8])
PY
timeout 300 python tools/gpu_stress.py 123 200 > gpurun_out/r3f2/stress.log 2>&1
grep -v "^slow" gpurun_out/r3f2/stress.log | tail -2
