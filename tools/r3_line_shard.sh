#!/bin/bash
# sharded line-metric path: tests, then timing at 16384 and 65536 over 8 in-process shards
mkdir -p gpurun_out/r3ls
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_line.py -x -q -m gpu > gpurun_out/r3ls/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r3ls/tests.log
tail -5 gpurun_out/r3ls/tests.log
timeout 300 python tools/r3_line_shard_time.py 16384 8 > gpurun_out/r3ls/line_shard_16384.json 2> gpurun_out/r3ls/line_shard_16384.err
tail -3 gpurun_out/r3ls/line_shard_16384.err
timeout 600 python tools/r3_line_shard_time.py 65536 8 > gpurun_out/r3ls/line_shard_65536.json 2> gpurun_out/r3ls/line_shard_65536.err
tail -3 gpurun_out/r3ls/line_shard_65536.err
cat gpurun_out/r3ls/line_shard_65536.json | tail -30
