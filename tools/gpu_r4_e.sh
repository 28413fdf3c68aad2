#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
bash tools/trace_workload.sh e_nest nest > gpurun_out/e_nest.log 2>&1; echo "nest trace rc=$?"
cd $R; python tools/kstats.py gpurun_out/e_nest/kernel_stats.csv 4 8
