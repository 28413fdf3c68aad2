#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/o_prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/o_prof -o o -- python3 tools/time_sub.py 2e7 > gpurun_out/o_prof.log 2>&1; echo "prof rc=$?"
CSV=$(find gpurun_out/o_prof -name "*kernel_stats.csv" | head -1); python tools/kstats.py $CSV 2 14
find gpurun_out/o_prof -name "*kernel_trace*" -delete
