#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { echo "$1: $(env $1 python3 bench.py --workload g1 --n 65536 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d.get('total_cost'), d['kernels'].get('sap'))")"; }
run "TD_SAPX_SLIM=4096"; run "TD_SAPX_SLIM=100000000"; run "TD_SAPX_SLIM=4096"
python3 tools/r3_shard_time.py 65536 8 gen 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('finisher', d['per_rank_ms']['finisher on rank 0'], d['projection'])"
python3 tools/r3_shard_time.py 65536 8 cost > gpurun_out/shard_time_cost.json 2>/dev/null; python3 -c "import json; d=json.load(open('gpurun_out/shard_time_cost.json')); print(json.dumps(d['per_rank_ms'])); print(d['projection']); print(d['total'], d['dual'], d['bytes_per_cell'])"
