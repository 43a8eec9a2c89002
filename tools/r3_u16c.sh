#!/bin/bash
mkdir -p gpurun_out/r3u16
for k in "geo2 10950" "geo2 5087" "geo2 14000" "mid 12000" "neg 9000" "wide 9000"; do
  for w in 0 32; do
    GEO_SIDE=4000 TD_U16_REDO_FREE=$w timeout 200 python3 tools/gpu_one.py $k 2 2>&1 | grep -v "amdgpu.ids" | tail -1 | cut -c1-300
  done
done > gpurun_out/r3u16/sweep3.log 2>&1
cat gpurun_out/r3u16/sweep3.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3u16/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3u16/tests.log
tail -3 gpurun_out/r3u16/tests.log
