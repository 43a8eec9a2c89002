#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4l
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --n ${N:-16384} --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4l/sw.json 2> gpurun_out/r4l/sw.err
  echo "$label: $(python -c "import json; d=json.loads(open('gpurun_out/r4l/sw.json').read().strip().splitlines()[-1]); print('%.1f M/s  %.4f ms/step  compress %.1f us  frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))")"
}
run "pipelined (default)" TD_LAZY_PIPE=1
run "plain lazy         " TD_LAZY_PIPE=0
run "pipelined grid=1   " TD_LAZY_PIPE=1 TD_BID0_GRID=1
run "pipelined grid=3   " TD_LAZY_PIPE=1 TD_BID0_GRID=3
N=12288 run "n=12288 pipelined" TD_LAZY_PIPE=1
N=12288 run "n=12288 plain lazy" TD_LAZY_PIPE=0
N=12288 run "n=12288 full copy " TD_LAZY_CC=0
