#!/bin/bash
# dev tool: state of the general solver at the start of round 3 (GPU box)
cd "$GRAFT_REPO_ROOT"
export TD_LINE=0
O=gpurun_out/r3base; mkdir -p $O
for k in "g2 16384" "geo2 16384" "wide 16384" "mid 16384" "g2 8192" "g2 4096"; do
  PROF=1 TD_DEBUG=1 python3 tools/gpu_one.py $k 2 >> $O/one.log 2>&1
done
tail -60 $O/one.log
