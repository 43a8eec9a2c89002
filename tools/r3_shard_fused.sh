#!/bin/bash
mkdir -p gpurun_out/r3sf
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3sf/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3sf/tests.log
tail -4 gpurun_out/r3sf/tests.log
for f in 1 0; do
TD_SHARD_FUSED0=$f timeout 600 python tools/r3_shard_time.py 65536 8 gen > gpurun_out/r3sf/shard_gen_fused$f.json 2> gpurun_out/r3sf/err$f.log
python - <<PY
import json
d=json.load(open("gpurun_out/r3sf/shard_gen_fused$f.json"))
p=d["per_rank_ms"]
print("fused0=$f", d["total"], d["dual"], "compress", round(p["compress (max over shards)"],3), "bid", p["bid per round (max over shards)"][:3], d["projection"])
PY
done
