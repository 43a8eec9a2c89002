#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/core_d.log
: > $L
for mult in 0.25 1 4 16; do
for kind in wide geo2 g2 mid; do
  for n in 16384; do
    TD_LINE=0 TD_DEBUG=1 TD_CORE_EPS=$mult timeout 900 python tools/gpu_one.py $kind $n 2 2>&1 | grep -e "warm start" -e "n=$n \[" | tail -2 | cut -c1-230 >> $L
  done
done
done
cat $L
