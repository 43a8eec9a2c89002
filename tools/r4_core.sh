#!/bin/bash
# round 4: the warm start on a sparse core (td_core_warm.h) — general solver on the tie-free families, with / without
mkdir -p gpurun_out/r4
L=gpurun_out/r4/core_${1:-a}.log
: > $L
for kind in wide geo2 g2 mid; do
  for n in 4096 16384; do
    for core in 0 1; do
      TD_LINE=0 TD_CORE=$core PROF=1 timeout 600 python tools/gpu_one.py $kind $n 2 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-330 >> $L
    done
  done
done
cat $L
