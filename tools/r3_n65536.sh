#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r3_n65536; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload g1 --n 65536 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
python3 bench.py --workload g1 --n 65536 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
head -28 $OUT/summary.txt; tail -2 $OUT/summary.txt | cut -c1-200
