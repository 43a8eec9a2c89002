#!/bin/bash
# round 4: where the eps schedule of the warm start begins (TD_WARM_DIV: eps0 = range / div), the four tie-free families
mkdir -p gpurun_out/r4
L=gpurun_out/r4/warmdiv.log
: > $L
for div in 4 32 256 2048; do
  for kind in wide mid geo2 g2; do
    TD_LINE=0 TD_WARM_DIV=$div PROF=1 timeout 600 python tools/gpu_one.py $kind 16384 2 2>&1 | grep -e " bid " -e " sap " -e "n=16384" | cut -c1-175 | tr '\n' ' ' >> $L; echo >> $L
  done
done
cat $L
