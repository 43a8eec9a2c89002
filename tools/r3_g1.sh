#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { echo "$1: $(env $1 python3 bench.py --workload g1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d.get('total_cost'))")"; }
run "TD_ROW_LOOP=0"; run "TD_ROW_LOOP=3"; run "TD_ROW_LOOP=5"; run "TD_ROW_LOOP=7"; run "TD_ROW_LOOP=99"; run "TD_ROW_LOOP=0"; run "TD_ROW_LOOP=99"; run "TD_ROW_LOOP=5"
