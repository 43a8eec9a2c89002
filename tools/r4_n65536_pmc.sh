#!/bin/bash
# N = 65 536 on one GPU: kernel trace + separate FETCH_SIZE / WRITE_SIZE passes (VERDICT r3 item 4 asks for all three)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r4_n65536
mkdir -p $OUT
ARGS="--n 65536 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS --steps 3 --warmup 1 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
python3 bench.py $ARGS --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | cut -c1-400
