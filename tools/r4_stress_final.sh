#!/bin/bash
# round 4: the four stress tools on the final binary, a few minutes each (0 failures expected)
mkdir -p gpurun_out/r4
L=gpurun_out/r4/stress_final.log
: > $L
timeout 400 python tools/gpu_stress.py 4101 240 2>&1 | grep -e "^stress" -e FAIL | head -20 >> $L
timeout 400 python tools/gpu_stress_large.py 4102 240 2>&1 | grep -e "large stress" -e FAIL | head -20 >> $L
STRESS_ALIGN=128 timeout 400 python tools/gpu_stress_bid0.py 4103 200 2>&1 | grep -e "stress bid0" -e FAIL | head -20 >> $L
TD_BLOCKS=0 timeout 300 python tools/gpu_stress_bid0.py 4104 120 2>&1 | grep -e "stress bid0" -e FAIL | head -20 >> $L
timeout 400 python tools/gpu_stress_tick.py 4105 240 2>&1 | grep -e "stress tick" -e FAIL | head -20 >> $L
cat $L
