#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/warmdiv2.log
: > $L
for kind in wide mid; do
for div in 2048 8192 32768; do
  for mr in 12 48; do
    TD_LINE=0 TD_WARM_DIV=$div TD_MAX_ROUNDS=$mr PROF=1 timeout 600 python tools/gpu_one.py $kind 16384 2 2>&1 | grep -e " bid " -e " sap " -e "n=16384" | cut -c1-190 | tr '\n' ' ' >> $L; echo >> $L
  done
done
done
for th in 4 8; do TD_LINE=0 TD_WARM_DIV=2048 TD_WARM_THETA=$th PROF=1 timeout 600 python tools/gpu_one.py wide 16384 2 2>&1 | grep -e " bid " -e " sap " -e "n=16384" | cut -c1-190 | tr '\n' ' ' >> $L; echo >> $L; done
cat $L
