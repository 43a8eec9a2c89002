#!/bin/bash
mkdir -p gpurun_out/r3last
python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3last/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r3last/tests.log
tail -3 gpurun_out/r3last/tests.log
timeout 600 python tools/r3_bid0_check.py 2>&1 | tail -1
timeout 200 python tools/gpu_stress.py 31 120 2>&1 | grep -v "^slow" | tail -1
timeout 200 python tools/gpu_stress_large.py 33 120 2>&1 | tail -1
timeout 200 python tools/gpu_stress_tick.py 35 120 2>&1 | tail -1
timeout 200 python tools/gpu_stress_bid0.py 37 100 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
