#!/bin/bash
mkdir -p gpurun_out/r3df
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r3df/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r3df/tests.log
tail -5 gpurun_out/r3df/tests.log
timeout 400 python tools/r3_shard_time.py 16384 8 padded > gpurun_out/r3df/padded_defer.json 2> gpurun_out/r3df/padded_defer.err
TD_DEFER_CONST=0 timeout 400 python tools/r3_shard_time.py 16384 8 padded > gpurun_out/r3df/padded_nodefer.json 2> gpurun_out/r3df/padded_nodefer.err
for f in padded_defer padded_nodefer; do echo $f; python - <<PY
import json
d=json.load(open("gpurun_out/r3df/$f.json"))
print(d["total"], d["dual"], d["per_rank_ms"]["bid per round (max over shards)"], d["per_rank_ms"]["finisher on rank 0"], d["projection"])
PY
done
