#!/bin/bash
mkdir -p gpurun_out/r4l
for n in 16384 12288 20480 32768; do
for cfg in "2 4" "2 5" "2 6"; do set -- $cfg
  TD_HOP_PASSES=$1 TD_ZS_ROUNDS=$2 timeout 300 python tools/r4_hop_seeds.py $n 10 2>&1 | tail -1 | cut -c1-120
done; done
for cfg in "2 4" "2 5" "2 6"; do set -- $cfg
  TD_HOP_PASSES=$1 TD_ZS_ROUNDS=$2 timeout 300 python tools/r4_hop_seeds.py 65536 3 2>&1 | tail -1 | cut -c1-120
  TD_HOP_PASSES=$1 TD_ZS_ROUNDS=$2 timeout 600 python tools/r4_shard_time.py 65536 8 gen 1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('   shards gen:', d['kernel_ms_per_step_on_8_gpus'], d['per_rank_kernel_ms (max over shards)'], 'left', d['rows_left_after_phase_a'])"
done
