#!/bin/bash
# GPU box: rocprofv3 kernel trace of the general solver on the |a-b| instance of bench.py (line metric off)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r3_g2gen
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload g2 --line-metric off --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
python3 bench.py --workload g2 --line-metric off --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err
PMC=0 python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
head -30 $OUT/summary.txt; tail -3 $OUT/bench.json | cut -c1-600
export TD_LINE=0
for k in "g2 16384" "geo2 16384" "wide 16384" "mid 16384" "g2 8192" "geo2 8192" "wide 8192" "g2 4096" "geo2 4096" "wide 4096"; do
  PROF=1 TD_DEBUG=1 python3 tools/gpu_one.py $k 3 2>&1 | grep -v "amdgpu.ids\|progress per" | tail -8 >> $OUT/one.log
done
grep "cert=" $OUT/one.log | cut -c1-150
