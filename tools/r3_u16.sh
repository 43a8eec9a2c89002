#!/bin/bash
mkdir -p gpurun_out/r3u16
export TD_LINE=0
for k in "geo2 2500" "geo2 3500" "geo2 5000" "geo2 7000" "mid 3000" "mid 5000" "mid 7000" "neg 5000"; do
  for w in 4096 2048 0; do
    TD_WIDE_U16_N=$w timeout 120 python3 tools/gpu_one.py $k 3 2>&1 | grep -v "amdgpu.ids" | tail -1 | cut -c1-330
  done
done > gpurun_out/r3u16/sweep.log 2>&1
cat gpurun_out/r3u16/sweep.log
