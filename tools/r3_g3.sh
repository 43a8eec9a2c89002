#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for w in g3 tick g1; do
  python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d.get('total_cost'), d.get('solver_stats'))"
done
TD_DEBUG=1 python3 tools/gpu_one.py g3 16384 1 2>&1 | grep -v amdgpu | tail -4
for k in "g2 16384" "wide 16384"; do TD_LINE=0 python3 tools/gpu_one.py $k 2 2>&1 | grep cert= | cut -c1-100; done
