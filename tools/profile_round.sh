#!/bin/bash
# Runs on the GPU box: kernel trace + two separate PMC passes of the default bench command.
# usage: tools/profile_round.sh <tag>
set -o pipefail
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
WL=${2:-g1}
EXTRA="--workload $WL --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline $EXTRA > $OUT/trace.log 2>&1
if [ "${PMC:-1}" = "1" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/pmc_write.log 2>&1
fi
python3 bench.py --steps ${STEPS:-20} --warmup 3 $EXTRA > $OUT/bench.json 2> $OUT/bench.err
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
