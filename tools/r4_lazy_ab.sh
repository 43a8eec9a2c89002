#!/bin/bash
# every workload of the default bench run with the lazy narrow copy on / off
set -o pipefail
mkdir -p gpurun_out/r4l
for lz in 1 0; do
  TD_LAZY_CC=$lz timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r4l/bench_default_lazy$lz.json 2> gpurun_out/r4l/bench_default_lazy$lz.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/r4l/bench_default_lazy$lz.json').read().strip().splitlines()[-1])
print('lazy=$lz', d['value'], d['ms_per_step'], d['roofline']['frac'])
for w in d.get('other_workloads', []):
    print('   ', {k: w[k] for k in w if k in ('workload', 'name', 'n', 'ms_per_step', 'ms', 'value')})
PY
done
for b in gen cost padded; do
  timeout 600 python tools/r4_shard_time.py 65536 8 $b 1 > gpurun_out/r4l/shard_time_${b}_blocks.json 2> gpurun_out/r4l/shard_time_${b}.err
  python -c "import json; d=json.load(open('gpurun_out/r4l/shard_time_${b}_blocks.json')); print('$b', {k: d[k] for k in d if 'ms' in k or 'left' in k or 'exch' in k})" | cut -c1-600
done
