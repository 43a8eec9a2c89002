#!/bin/bash
mkdir -p gpurun_out/r3st
timeout 500 python tools/gpu_stress_tick.py 1 360 > gpurun_out/r3st/tick.log 2>&1
tail -8 gpurun_out/r3st/tick.log
timeout 400 python tools/gpu_stress.py 77 300 > gpurun_out/r3st/assign.log 2>&1
grep -v "^slow" gpurun_out/r3st/assign.log | tail -5
