#!/bin/bash
# round 4: after the k_hop_table / k_hop_match rewrite — correctness slice, timing, kernel trace of the g1 step
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/r4_blocks.py 16384 12288 2>&1 | grep -v amdgpu.ids
for n in 16384 65536; do TD_BLOCKS=8 timeout 300 python tools/gpu_one.py g1 $n 7 2>&1 | tail -1 | cut -c1-90; done
PMC=0 STEPS=10 bash tools/profile_round.sh r4b g1 2>&1 | tail -32
timeout 600 python tools/r4_shard_time.py 65536 8 gen 1 > gpurun_out/r4/shard_time_n65536_8shards_gen_blocks_b.json 2> gpurun_out/r4/shard_time_gen_b.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4/shard_time_n65536_8shards_gen_blocks_b.json"))
print(d["sequence"], "left", d["rows_left_after_phase_a"], "kernel ms/rank", d["kernel_ms_per_step_on_8_gpus"], d["per_rank_kernel_ms (max over shards)"])
PY
