#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4l
run() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --n ${N:-16384} --steps ${ST:-30} --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4l/sw.json 2> gpurun_out/r4l/sw.err
  echo "$label: $(python -c "import json; d=json.loads(open('gpurun_out/r4l/sw.json').read().strip().splitlines()[-1]); print('%.2f M/s  %.4f ms/step  compress %.1f us  frac %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac']))")"
}
for rep in 1 2; do
run "n=16384 compile-time diag" TD_LAZY_DIAG=1
run "n=16384 runtime diag     " TD_LAZY_DIAG=0
done
N=12288 run "n=12288 compile-time diag" TD_LAZY_DIAG=1
N=12288 run "n=12288 runtime diag     " TD_LAZY_DIAG=0
N=65536 ST=5 run "n=65536 compile-time diag" TD_LAZY_DIAG=1
N=65536 ST=5 run "n=65536 runtime diag     " TD_LAZY_DIAG=0
