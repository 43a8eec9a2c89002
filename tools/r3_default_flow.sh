#!/bin/bash
mkdir -p gpurun_out/r3df2
for k in "geo2 4096" "geo2 8192" "geo2 16384" "wide 16384" "mid 16384" "wide 4096" "neg 16384"; do
  GEO_SIDE=4000 PROF=1 timeout 300 python3 tools/gpu_one.py $k 3 2>&1 | grep -v "amdgpu.ids" | tail -7 | cut -c1-300
done > gpurun_out/r3df2/default_flow.log 2>&1
cat gpurun_out/r3df2/default_flow.log
