#!/bin/bash
mkdir -p gpurun_out/r3fc
timeout 900 python tools/r3_bid0_check.py > gpurun_out/r3fc/check.log 2>&1; echo "check rc $?" >> gpurun_out/r3fc/check.log
tail -3 gpurun_out/r3fc/check.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3fc/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3fc/tests.log
tail -3 gpurun_out/r3fc/tests.log
timeout 400 python tools/gpu_stress_bid0.py 3 300 > gpurun_out/r3fc/stress_bid0.log 2>&1
tail -3 gpurun_out/r3fc/stress_bid0.log
timeout 300 python tools/gpu_stress.py 91 200 > gpurun_out/r3fc/stress.log 2>&1
grep -v "^slow" gpurun_out/r3fc/stress.log | tail -3
timeout 300 python tools/gpu_stress_large.py 17 200 > gpurun_out/r3fc/stress_large.log 2>&1
tail -3 gpurun_out/r3fc/stress_large.log
timeout 600 python bench.py > gpurun_out/r3fc/bench_default.json 2> gpurun_out/r3fc/bench_default.err
tail -c 1500 gpurun_out/r3fc/bench_default.json | cut -c1-800
