#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r3_tick
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload tick --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
python3 bench.py --workload tick --steps 300 --warmup 30 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
head -40 $OUT/summary.txt
