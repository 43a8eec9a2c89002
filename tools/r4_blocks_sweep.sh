#!/bin/bash
# round 4: the block-local start on the perf.jl instance — block count, local rounds, sizes (one GPU)
mkdir -p gpurun_out/r4
L=gpurun_out/r4/blocks_sweep.log
: > $L
TD_DEBUG=1 TD_BLOCKS=8 timeout 300 python tools/gpu_one.py g1 16384 1 2>&1 | grep -v "^\[td\] n=" | tail -8 >> $L
for n in 16384 65536; do
  for b in 0 1 4 8 16; do
    TD_BLOCKS=$b timeout 300 python tools/gpu_one.py g1 $n 5 2>&1 | tail -1 >> $L
  done
  for zr in 1 2 3 5 6; do
    TD_BLOCKS=8 TD_ZS_ROUNDS=$zr timeout 300 python tools/gpu_one.py g1 $n 5 2>&1 | tail -1 >> $L
  done
done
for n in 12288 20480 32768; do
  for b in 0 8; do
    TD_BLOCKS=$b timeout 300 python tools/gpu_one.py g1 $n 5 2>&1 | tail -1 >> $L
  done
done
cat $L
