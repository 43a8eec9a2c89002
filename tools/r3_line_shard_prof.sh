#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ls; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_ls
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ls -o ls -- python3 tools/r3_line_shard_time.py 65536 8 > gpurun_out/prof_ls/run.json 2> gpurun_out/prof_ls/run.err
echo "rc $?"
find gpurun_out/prof_ls -name "*kernel_stats.csv" | while read f; do head -14 "$f"; done
find gpurun_out/prof_ls -name "*kernel_trace.csv" -size +20M -delete
