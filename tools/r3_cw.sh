#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TD_LINE=0
run() { echo "$1 | $2: $(env $1 TD_DEBUG=1 timeout 120 python3 tools/gpu_one.py $2 2 2>&1 | grep 'k_forest\|cert=' | tail -2 | sed 's/.*levels/levels/; s/\[TD.*\]//; s/total=.*cert/cert/' | cut -c1-330 | tr '\n' ' ' | sed "s/{'bid_rounds': 12, //; s/'sap_free.*//")"; }
for k in "g2 4096" "g2 16384" "geo2 16384" "wide 16384" "mid 16384"; do run "A=1" "$k"; done
for k in "g2 16384" "geo2 16384" "wide 16384"; do
  run "TD_WARM_THETA=4" "$k"; run "TD_WARM_BITS=8" "$k"; run "TD_WARM_BITS=12 TD_WARM_THETA=4" "$k"; run "TD_WARM_CUT=16" "$k"
done
