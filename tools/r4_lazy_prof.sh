#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4l
TD_LAZY_CC=0 STEPS=10 PMC=0 bash tools/profile_round.sh r4l_lazy0 g1 > /dev/null 2>&1
TD_LAZY_CC=1 STEPS=10 PMC=0 bash tools/profile_round.sh r4l_lazy1 g1 > /dev/null 2>&1
for t in r4l_lazy0 r4l_lazy1; do echo "== $t"; head -20 gpurun_out/prof_$t/summary.txt; done
