#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/general_solver_r4_schedule.txt
: > $L
for kind in wide mid neg geo2 g2; do
  for n in 8192 16384; do
    TD_LINE=0 TD_DEBUG=1 timeout 900 python tools/gpu_one.py $kind $n 2 2>&1 | grep -e "row-correlation" -e "n=$n \[" | tail -2 | cut -c1-215 >> $L
  done
done
cat $L
