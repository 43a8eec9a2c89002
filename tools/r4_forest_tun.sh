#!/bin/bash
# runtime tunables of the forest finisher on the three tie-free families (no binary change)
cd "$GRAFT_REPO_ROOT"
export TD_LINE=0
run() { echo "$1 | $2: $(env $1 timeout 200 python3 tools/gpu_one.py $2 3 2>&1 | grep 'cert=' | tail -1 | sed 's/\[TD.*\]//; s/total=.*warm_rounds/warm_rounds/; s/bytes_per_cell.*forest_levels/forest_levels/' | cut -c1-200)"; }
for e in "A=1" "TD_FOREST_W0=64" "TD_FOREST_W0=256" "TD_FOREST_WX=64" "TD_FOREST_WX=256" "TD_FOREST_WX=0" "TD_FOREST_TB=512" "TD_FOREST_TB=256" "TD_FOREST_CW=128" "TD_FOREST_CW=128 TD_FOREST_TB=256"; do
  for k in "wide 16384" "geo2 16384"; do run "$e" "$k"; done
done
