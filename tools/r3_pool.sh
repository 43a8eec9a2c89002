#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r3_pool; rm -rf $OUT; mkdir -p $OUT
python3 bench.py --workload pool --steps 5 > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/prof_r3_pool/bench.json').read().strip().splitlines()[-1])
for k,v in d['pool'].items(): print(k, {f:(round(x,3) if isinstance(x,float) else x) for f,x in v.items()})
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload pool --steps 2 --no-cpu-baseline > $OUT/trace.log 2>&1
python3 tools/summarize_profile.py $OUT 2>/dev/null | head -16
