#!/bin/bash
# usage: r3_profile_wl.sh <workload> <tag> [steps]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
WL=$1; OUT=gpurun_out/prof_r3_$2; ST=${3:-20}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $WL --steps $ST --warmup 3 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
if [ "${PMC:-0}" = "1" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
fi
python3 bench.py --workload $WL --steps $((ST*3)) --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
head -${LINES_OUT:-30} $OUT/summary.txt; tail -4 $OUT/summary.txt | cut -c1-300
