#!/bin/bash
mkdir -p gpurun_out/r4l
for n in 16384 12288 20480; do
for cfg in "2 4" "1 4" "1 5" "1 6" "1 8"; do set -- $cfg
  TD_HOP_PASSES=$1 TD_ZS_ROUNDS=$2 timeout 300 python tools/r4_hop_seeds.py $n 10 2>&1 | tail -1
done; done
for cfg in "2 4" "1 4" "1 6"; do set -- $cfg
  TD_HOP_PASSES=$1 TD_ZS_ROUNDS=$2 timeout 300 python tools/r4_hop_seeds.py 65536 2 2>&1 | tail -1
done
