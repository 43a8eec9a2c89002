#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4l/rest
TD_LAZY_CC=1 python3 tools/r4_rest_time.py 16384 | tail -1 | cut -c1-300
TD_LAZY_CC=0 python3 tools/r4_rest_time.py 16384 | tail -1 | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4l/rest -- python3 tools/r4_rest_time.py 16384 > gpurun_out/r4l/rest.log 2>&1
f=$(ls gpurun_out/r4l/rest/*/*kernel_stats.csv | head -1)
head -12 $f | cut -c1-160
