#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4l
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --n 16384 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4l/sw.json 2> gpurun_out/r4l/sw.err
  echo "$label: $(python -c "import json; d=json.loads(open('gpurun_out/r4l/sw.json').read().strip().splitlines()[-1]); print('%.1f M/s  %.4f ms/step  compress %.1f us  frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))")"
}
run "wave-per-row default (wpc 16, u 8)" TD_LAZY_CC=1
run "workgroup-per-row lazy           " TD_LAZY_CC=1 TD_DIAG_WPC=0
run "workgroup-per-row lazy grid=1    " TD_LAZY_CC=1 TD_DIAG_WPC=0 TD_BID0_GRID=1
for u in 4 8 16; do for w in 8 12 16 24 32; do run "wave-per-row wpc=$w u=$u" TD_DIAG_WPC=$w TD_DIAG_U=$u; done; done
