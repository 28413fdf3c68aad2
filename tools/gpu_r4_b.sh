#!/bin/bash
# round 4, second GPU session: overlapped passes + traces of config 5 and of a shard-sized graph
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/b_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/b_tests.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/b_bench1.json 2> gpurun_out/b_bench1.err; echo "bench rc=$?"
tail -c 400 gpurun_out/b_bench1.err
bash tools/trace_workload.sh b_nest nest > gpurun_out/b_nest.log 2>&1; echo "nest trace rc=$?"
cd $R
bash tools/trace_workload.sh b_shard hprc-wg --scale 0.125 > gpurun_out/b_shard.log 2>&1; echo "shard trace rc=$?"
cd $R
python tools/trace_timeline.py $(ls gpurun_out/b_shard/kt/*/*kernel_trace.csv 2>/dev/null | tail -1) > gpurun_out/b_shard_timeline.txt 2>&1
