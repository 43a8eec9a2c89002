#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/core_e.log
: > $L
for pat in 3 6; do
for kind in wide geo2 g2 mid; do
  for n in 4096 16384; do
    TD_LINE=0 TD_DEBUG=1 TD_CORE_PATIENCE=$pat timeout 900 python tools/gpu_one.py $kind $n 3 2>&1 | grep -e "warm start" -e "n=$n \[" | tail -2 | cut -c1-200 >> $L
  done
done
done
for kind in wide mid; do TD_LINE=0 TD_CORE=0 timeout 900 python tools/gpu_one.py $kind 16384 3 2>&1 | tail -1 | cut -c1-200 >> $L; done
cat $L
