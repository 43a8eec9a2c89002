#!/bin/bash
# round 4: block-local start + global two-hop pass + early read-back: timing, debug, exactness stress with 8 blocks forced
mkdir -p gpurun_out/r4
L=gpurun_out/r4/blocks2.log
: > $L
TD_DEBUG=1 TD_BLOCKS=8 timeout 300 python tools/gpu_one.py g1 16384 1 2>&1 | grep -v "^\[td\] n=" | tail -8 >> $L
for n in 12288 16384 32768 65536; do
  for b in 0 4 8 16; do
    TD_BLOCKS=$b timeout 300 python tools/gpu_one.py g1 $n 7 2>&1 | tail -1 >> $L
  done
done
timeout 300 python tools/r4_blocks.py 16384 12288 >> $L 2>&1
STRESS_ALIGN=128 TD_BLOCKS=8 timeout 400 python tools/gpu_stress_bid0.py 3 240 >> $L 2>&1
TD_BLOCKS=8 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5 >> $L
cat $L
