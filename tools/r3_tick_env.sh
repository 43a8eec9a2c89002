#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { echo "$1: $(env $1 python3 bench.py --workload tick --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d.get('total_cost'))")"; }
run "A=1"; run "TD_PSAP_MIN=32"; run "TD_PSAP_MIN=64"; run "TD_FUSED_ROUNDS=6"; run "TD_FUSED_ROUNDS=5 TD_PSAP_MIN=32"; run "TD_FUSE_T=1"; run "TD_FUSE_T=0"; run "A=2"
