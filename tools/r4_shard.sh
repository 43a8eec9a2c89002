#!/bin/bash
# round 4: sharded solve, new sequence — GPU tests of the shard paths, then the per-rank timing at N = 65 536 / 8 shards
mkdir -p gpurun_out/r4 profiles/r4
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -8
for b in gen cost padded; do
  timeout 600 python tools/r4_shard_time.py 65536 8 $b 1 > gpurun_out/r4/shard_time_n65536_8shards_${b}_blocks.json 2> gpurun_out/r4/shard_time_${b}.err || tail -5 gpurun_out/r4/shard_time_${b}.err
done
timeout 600 python tools/r4_shard_time.py 65536 8 gen 0 > gpurun_out/r4/shard_time_n65536_8shards_gen_plain.json 2>> gpurun_out/r4/shard_time_gen.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4/shard_time_n65536_8shards_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["sequence"], "left", d["rows_left_after_phase_a"], "kernel ms/rank", d["kernel_ms_per_step_on_8_gpus"])
    print("   per rank", d["per_rank_kernel_ms (max over shards)"])
    print("   replicated", d["replicated_on_every_rank_ms"], "rank0", d["rank0_only_ms"])
PY
