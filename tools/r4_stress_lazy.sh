#!/bin/bash
# the families that reach the lazy narrow copy (1-byte rows, n >= 12 288, 128-aligned sizes: block-local start), longer
mkdir -p gpurun_out/r4
L=gpurun_out/r4/stress_lazy.log
: > $L
STRESS_BOTH=1 STRESS_ALIGN=128 timeout 700 python tools/gpu_stress_bid0.py 5101 600 2>&1 | grep -e "stress bid0" -e FAIL | head -20 >> $L
STRESS_ALIGN=128 TD_LAZY_CC=0 timeout 400 python tools/gpu_stress_bid0.py 5101 300 2>&1 | grep -e "stress bid0" -e FAIL | head -20 >> $L
timeout 500 python tools/gpu_stress_tick.py 5102 400 2>&1 | grep -e "stress tick" -e FAIL | head -20 >> $L
cat $L
