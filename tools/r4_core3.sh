#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/core_c.log
: > $L
for kind in wide geo2 g2 mid; do
  for n in 4096 16384; do
    TD_LINE=0 TD_DEBUG=1 timeout 900 python tools/gpu_one.py $kind $n 2 2>&1 | grep -e "warm start" -e "n=$n \[" | tail -2 | cut -c1-230 >> $L
  done
done
echo "--- schedule sweep, uniform 0..1e6 n=16384" >> $L
for cut in 64 256 1024; do
  for k in 64 128; do
    TD_LINE=0 TD_WARM_CUT=$cut TD_CORE_K=$k PROF=1 timeout 300 python tools/gpu_one.py wide 16384 2 2>&1 | grep -e " bid " -e " sap " -e "n=16384" | cut -c1-200 >> $L
  done
done
cat $L
