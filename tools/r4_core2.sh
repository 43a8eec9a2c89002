#!/bin/bash
# round 4: warm start on the sparse core with the dense fall-back for flagged rows; schedule sweep on the uniform family
mkdir -p gpurun_out/r4
L=gpurun_out/r4/core_b.log
: > $L
for kind in wide geo2 g2 mid; do
  for n in 4096 16384; do
    TD_LINE=0 TD_DEBUG=1 timeout 900 python tools/gpu_one.py $kind $n 2 2>&1 | grep -e "warm start" -e "n=$n \[" | tail -2 | cut -c1-300 >> $L
  done
done
echo "--- schedule sweep, uniform 0..1e6 n=16384" >> $L
for cut in 64 256 1024 4096; do
  for k in 32 64 100; do
    TD_LINE=0 TD_WARM_CUT=$cut TD_CORE_K=$k timeout 300 python tools/gpu_one.py wide 16384 2 2>&1 | tail -1 | cut -c1-250 >> $L
  done
done
for theta in 2 4; do for div in 4 16 64; do
  TD_LINE=0 TD_WARM_CUT=1024 TD_WARM_THETA=$theta TD_WARM_DIV=$div timeout 300 python tools/gpu_one.py wide 16384 2 2>&1 | tail -1 | cut -c1-250 >> $L
done; done
cat $L
