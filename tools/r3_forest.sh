#!/bin/bash
# dev tool: k_forest against the old finishers (GPU box)
cd "$GRAFT_REPO_ROOT"
export TD_LINE=0
O=gpurun_out/r3forest; mkdir -p $O; rm -f $O/one.log
for k in "g2 4096" "geo2 4096" "g2 6000" "wide 16384" "g2 16384" "geo2 16384" "mid 16384" "g2 20000"; do
  TD_DEBUG=1 timeout 120 python3 tools/gpu_one.py $k 2 >> $O/one.log 2>&1 || echo "FAILED/TIMEOUT $k rc=$?" >> $O/one.log
done
grep -v "amdgpu.ids\|progress per round" $O/one.log | tail -70
