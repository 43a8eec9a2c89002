#!/bin/bash
mkdir -p gpurun_out/r3u16
export TD_LINE=0
for k in "geo2 6000" "geo2 10950" "geo2 14000" "mid 12000" "mid 6000" "neg 9000"; do
  for w in 0 32 128; do
    GEO_SIDE=4000 TD_U16_REDO_FREE=$w timeout 200 python3 tools/gpu_one.py $k 2 2>&1 | grep -v "amdgpu.ids" | tail -1 | cut -c1-300
  done
done > gpurun_out/r3u16/sweep2.log 2>&1
cat gpurun_out/r3u16/sweep2.log
