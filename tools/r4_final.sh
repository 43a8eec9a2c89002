#!/bin/bash
# round 4: final measurements with the final binary (one gpurun call)
set -o pipefail
mkdir -p gpurun_out/r4 profiles/r4
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r4/final_pytest_gpu.txt
cat gpurun_out/r4/final_pytest_gpu.txt
STEPS=10 PMC=1 bash tools/profile_round.sh r4 g1 > /dev/null 2>&1
cp gpurun_out/prof_r4/summary.json profiles/r4/rocprof_summary_r4_g1.json
STEPS=10 PMC=1 bash tools/profile_round.sh r4_g3 g3 > /dev/null 2>&1
STEPS=20 PMC=0 bash tools/profile_round.sh r4_tick tick > /dev/null 2>&1
mkdir -p gpurun_out/prof_r4_n65536
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r4_n65536/trace -- python3 bench.py --n 65536 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_r4_n65536/trace.log 2>&1
python3 bench.py --n 65536 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_r4_n65536/bench.json 2> gpurun_out/prof_r4_n65536/bench.err
python3 tools/summarize_profile.py gpurun_out/prof_r4_n65536 > gpurun_out/prof_r4_n65536/summary.txt 2>&1
for b in gen cost padded; do
  timeout 600 python tools/r4_shard_time.py 65536 8 $b 1 > gpurun_out/r4/shard_time_n65536_8shards_${b}_blocks.json 2> gpurun_out/r4/shard_time_${b}.err
done
timeout 600 python tools/r4_shard_time.py 65536 8 gen 0 > gpurun_out/r4/shard_time_n65536_8shards_gen_plain.json 2>> gpurun_out/r4/shard_time_gen.err
timeout 600 python bench.py --force-sharded --steps 10 --warmup 2 --no-cpu-baseline --no-extras --sharded-n 65536 --n 16384 > gpurun_out/r4/bench_r4_force_sharded_n65536_one_rank.json 2>/dev/null
timeout 900 python bench.py > gpurun_out/r4/bench_r4_default.json 2> gpurun_out/r4/bench_r4_default.err
tail -c 1500 gpurun_out/r4/bench_r4_default.json
for f in gpurun_out/prof_r4/summary.txt gpurun_out/prof_r4_g3/summary.txt gpurun_out/prof_r4_tick/summary.txt gpurun_out/prof_r4_n65536/summary.txt; do echo "== $f"; head -24 $f; done
