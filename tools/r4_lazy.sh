#!/bin/bash
# round 4: the lazy narrow copy (TD_LAZY_CC) against the full one — tests first, then g1 at n = 16 384 / 65 536 both ways
set -o pipefail
mkdir -p gpurun_out/r4l
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r4l/pytest_gpu.txt
cat gpurun_out/r4l/pytest_gpu.txt
for lz in 0 1; do
  for n in 16384 65536; do
    st=20; [ $n = 65536 ] && st=5
    TD_LAZY_CC=$lz timeout 600 python bench.py --n $n --steps $st --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4l/bench_n${n}_lazy${lz}.json 2> gpurun_out/r4l/bench_n${n}_lazy${lz}.err
    echo "lazy=$lz n=$n: $(python -c "import json,sys; d=json.loads(open('gpurun_out/r4l/bench_n${n}_lazy${lz}.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'))")"
  done
done
TD_DEBUG=1 timeout 300 python bench.py --n 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "\[td\]" | tail -8
