#!/bin/bash
# dev tool: k_forest window parameters (GPU box)
cd "$GRAFT_REPO_ROOT"
export TD_LINE=0
for wx in 16 256 4096 65536; do for w0 in 16 1024; do
for k in "wide 16384" "geo2 16384" "g2 16384" "mid 16384"; do
  echo "WX=$wx W0=$w0 $k: $(TD_FOREST_WX=$wx TD_FOREST_W0=$w0 TD_DEBUG=1 timeout 120 python3 tools/gpu_one.py $k 1 2>&1 | grep 'k_forest\|cert=' | tail -2 | sed 's/.*levels/levels/; s/| Mcycles.*//; s/\[TD.*\]//; s/total=.*cert/cert/' | cut -c1-120 | tr '\n' ' ')"
done; done; done
