#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TD_LINE=0
run() { echo "$1 | $2: $(env $1 timeout 120 python3 tools/gpu_one.py $2 3 2>&1 | grep 'cert=' | tail -1 | sed 's/\[TD.*\]//; s/total=.*warm_rounds/warm_rounds/; s/sap_steps.*forest_levels/forest_levels/' | cut -c1-160)"; }
for e in "A=1" "TD_WARM_GROUPS=4" "TD_WARM_GROUPS=2" "TD_WARM_DIV=16" "TD_WARM_DIV=16 TD_WARM_GROUPS=4" "TD_WARM_DIV=64 TD_WARM_GROUPS=4" "TD_WARM_GROUPS=4 TD_WARM_CUT=32"; do
  for k in "wide 16384" "geo2 16384" "g2 16384" "mid 16384"; do run "$e" "$k"; done
done
