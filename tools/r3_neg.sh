#!/bin/bash
mkdir -p gpurun_out/r3df2
for e in "TD_U16_REDO_FREE=0" "TD_U16_REDO_FREE=64" "TD_U16_REDO_FREE=512" "TD_WARM_TIE_DIV=1" "TD_WARM_TIE_DIV=1 TD_U16_REDO_FREE=64"; do
  for k in "neg 16384" "neg 12000"; do
    env $e timeout 300 python3 tools/gpu_one.py $k 2 2>&1 | grep -v "amdgpu.ids" | tail -1 | cut -c1-250
  done
done > gpurun_out/r3df2/neg.log 2>&1
cat gpurun_out/r3df2/neg.log
